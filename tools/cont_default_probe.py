"""The reference's DEFAULT gradient (ContinuousAdjoint, A(T) law) on the bench workload, G x n^2 (default 64 x 1024^2): wall clock, the host
phases (ODINN_PROFILE_HOST), and -- under rocprofv3 --kernel-trace --stats -- the kernel shares.  python tools/cont_default_probe.py [n G law]
law: A (default), Y (bench.json's bench_workload_Y_law) or U (bench_workload_U_law)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
gl = [make_glacier(n, k) for k in range(G)]
b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl])
for k, (H0, B, A) in enumerate(gl):
    b.set_fields(k, H0, B)
ph = odinn.PhysicalParameters()
nn = odinn.NeuralNetwork(odinn.Parameters(), seed=666)
mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA)
law = sys.argv[3] if len(sys.argv) > 3 else "A"
theta = nn.theta
if law == "A":
    b.set_law(odinn.LAW_NN_A_SCALAR, mlp, nn.theta)
elif law == "Y":
    m = odinn.MLPSpec([2, 3, 10, 3, 1], [odinn.ACT_SOFTPLUS] * 3 + [odinn.ACT_SIGMOID], [(-25.0, 0.0), (0.0, 500.0)], odinn.POST_EXPMAX, 0.0, ph.maxA)
    theta = np.random.default_rng(1234).uniform(-0.5, 0.5, m.n_params)
    b.set_law(odinn.LAW_NN_Y, m, theta)
else:
    m = odinn.MLPSpec([2, 3, 10, 3, 1], [odinn.ACT_SOFTPLUS] * 3 + [odinn.ACT_SIGMOID], [(0.0, 300.0), (0.0, 0.5)], odinn.POST_EXPMAX, 0.0, 50.0)
    theta = np.random.default_rng(1234).uniform(-0.5, 0.5, m.n_params)
    b.set_law(odinn.LAW_NN_U, m, theta)
ts = [2010.0 + k / 12.0 for k in range(25)]
for k in range(G):
    b.set_reference(k, ts, [gl[k][0] * (1.0 - 0.002 * j) for j in range(len(ts))], 3)
f = lambda: b.batch_loss_grad(None, ts, theta=theta, continuous=True, reltol=1e-8)
f(); b.sync()
os.environ["ODINN_PROFILE_HOST"] = "1"
t0 = time.perf_counter(); f(); b.sync(); t = time.perf_counter() - t0
rev = b.last_stats_rev[0]
print("continuous gradient of %d x %d^2: %.1f ms, %d+%d reverse steps, %.1f grad-evals/s" % (G, n, t * 1e3, rev.naccept, rev.nreject, G / t))
