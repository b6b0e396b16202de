"""Gradient of the bench job with the headline's gridded law A = NN_theta(T) (2x16 MLP, 321 parameters): where a discrete /
continuous gradient evaluation spends its time.  usage: grad_gridded_probe.py [G=64] [n=1024] [discrete|continuous|both]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier, temperature_field
G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
what = sys.argv[3] if len(sys.argv) > 3 else "both"
ph = odinn.PhysicalParameters()
gl = [make_glacier(n, k) for k in range(G)]
b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl])
for k, (H0, B, A) in enumerate(gl):
    b.set_fields(k, H0, B)
    b.set_T_field(k, temperature_field(H0, B))
mlpA = odinn.MLPSpec([1, 16, 16, 1], [odinn.ACT_SOFTPLUS, odinn.ACT_SOFTPLUS, odinn.ACT_SIGMOID], None, odinn.POST_AFFINE, ph.minA, ph.maxA)
thetaA = np.random.default_rng(1234).uniform(-0.5, 0.5, mlpA.n_params)
b.set_law(odinn.LAW_NN_A_GRIDDED, mlpA, thetaA)
ts = [2010.0 + k / 12.0 for k in range(25)]
for k in range(G):
    b.set_reference(k, ts, [gl[k][0] * (1.0 - 0.002 * j) for j in range(len(ts))], 3)
def tm(f, reps=2):
    f(); b.sync()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    b.sync()
    return (time.perf_counter() - t0) / reps * 1e3
st = b.solve(ts, reltol=1e-8)
print("solve ms", tm(lambda: b.solve(ts, reltol=1e-8)), sorted((s.naccept, s.nreject) for s in st)[::max(1, G // 4)])
if what in ("discrete", "both"):
    print("discrete ms", tm(lambda: b.batch_loss_grad(None, ts, theta=thetaA, continuous=False, reltol=1e-8)))
if what in ("continuous", "both"):
    print("continuous ms", tm(lambda: b.batch_loss_grad(None, ts, theta=thetaA, continuous=True, reltol=1e-8)), b.last_stats_rev[0].naccept, b.last_stats_rev[0].nreject)
b.close()
