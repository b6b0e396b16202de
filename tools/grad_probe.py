"""Where a grad-eval of bench.py's workload (8 x 1024^2, 3 monthly snapshots) spends its time."""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier
n, G = 1024, 8
gl = [make_glacier(n, k) for k in range(G)]
b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl])
for k, (H0, B, A) in enumerate(gl):
    b.set_fields(k, H0, B)
ph = odinn.PhysicalParameters()
nn = odinn.NeuralNetwork(odinn.Parameters(), seed=666)
mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA)
b.set_law(odinn.LAW_NN_A_SCALAR, mlp, nn.theta)
ts = [2010.0 + k / 12.0 for k in range(4)]
for k in range(G):
    b.set_reference(k, ts, [gl[k][0] * (1.0 - 0.01 * j) for j in range(len(ts))], 3)
def tm(f, n=10):
    f(); b.sync()
    t0 = time.perf_counter()
    for _ in range(n): f()
    b.sync()
    return (time.perf_counter() - t0) / n * 1e3
print("solve ms", tm(lambda: b.solve(ts, reltol=1e-6)), [ (s.naccept, s.nreject) for s in b.solve(ts, reltol=1e-6)][:2])
print("loss_grad ms", tm(lambda: b.loss_grad(ts, theta=nn.theta, reltol=1e-6)))
print("loss_grad (theta unchanged) ms", tm(lambda: b.loss_grad(ts, reltol=1e-6)))
os.environ["ODINN_PROFILE_HOST"] = "1"
for mode in ("0", "1"):
    os.environ["ODINN_SCHEDULE"] = "adj_fused=" + mode
    print("continuous loss_grad ms (ODINN_ADJ_FUSED=%s)" % mode, tm(lambda: b.loss_grad_continuous(ts, theta=nn.theta, reltol=1e-6), n=2), b.last_stats_rev[0].naccept, b.loss_grad_continuous(ts, theta=nn.theta, reltol=1e-6)[0])
