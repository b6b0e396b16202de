"""The bench line's Y-law discrete gradient (bench.py: bench_workload_Y_law) on G x n^2 glaciers: wall clock, forward step counts,
table rebuilds (run under ODINN_LAW_TABLE_VERBOSE=1 / rocprofv3 --kernel-trace --stats): python tools/ylaw_disc_probe.py [n G]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ph = odinn.PhysicalParameters()
gl = [make_glacier(n, j) for j in range(G)]
b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl])
for j, (H0, B, A) in enumerate(gl):
    b.set_fields(j, H0, B)
mY = odinn.MLPSpec([2, 3, 10, 3, 1], [odinn.ACT_SOFTPLUS] * 3 + [odinn.ACT_SIGMOID], [(-25.0, 0.0), (0.0, 500.0)], odinn.POST_EXPMAX, 0.0, ph.maxA)
thY = np.random.default_rng(1234).uniform(-0.5, 0.5, mY.n_params)
b.set_law(odinn.LAW_NN_Y, mY, thY)
ts = [2010.0 + k / 12.0 for k in range(25)]
for k in range(G):
    b.set_reference(k, ts, [gl[k][0] * (1.0 - 0.002 * j) for j in range(len(ts))], 3)
def tm(f, nrep=2):
    f(); b.sync()
    t0 = time.perf_counter()
    for _ in range(nrep): f()
    b.sync()
    return (time.perf_counter() - t0) / nrep * 1e3
st = b.solve(ts, reltol=1e-8)
print("solve ms %.2f" % tm(lambda: b.solve(ts, reltol=1e-8)), "forward steps (accepted, rejected) of the first glaciers:", [(s.naccept, s.nreject) for s in st][:4])
print("discrete ms %.2f" % tm(lambda: b.batch_loss_grad(None, ts, theta=thY, continuous=False, reltol=1e-8)), "table usable:", b.law_table()["usable"])
