"""Where a discrete-adjoint grad-eval of the bench workload (8 x 1024^2, k = 25, reltol 1e-8) spends its time."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
ts = [2010.0 + k / 12.0 for k in range(25)]
for k in range(G):
    b.set_reference(k, ts, [gl[k][0] * (1.0 - 0.002 * j) for j in range(len(ts))], 3)
def tm(f, n=3):
    f(); b.sync()
    t0 = time.perf_counter()
    for _ in range(n): f()
    b.sync()
    return (time.perf_counter() - t0) / n * 1e3
print("solve ms", tm(lambda: b.solve(ts, reltol=1e-8)), [(s.naccept, s.nreject) for s in b.solve(ts, reltol=1e-8)][:2])
print("solve dense ms", tm(lambda: b.solve(ts, reltol=1e-8, dense=1)))
print("loss_grad ms", tm(lambda: b.loss_grad(ts, theta=nn.theta, reltol=1e-8)))
print("loss_grad (theta unchanged) ms", tm(lambda: b.loss_grad(ts, reltol=1e-8)))
os.environ["ODINN_PROFILE_HOST"] = "1"
b.solve(ts, reltol=1e-8)
print("continuous ms", tm(lambda: b.loss_grad_continuous(ts, theta=nn.theta, reltol=1e-8), n=1), b.last_stats_rev[0].naccept, b.last_stats_rev[0].nreject)
