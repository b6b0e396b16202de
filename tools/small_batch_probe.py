"""configs[3] literally: 4 alpine glaciers, k = 25, reltol 1e-8 -- wall time of solve / discrete / continuous gradient
(run under rocprofv3 --kernel-trace --stats to see the kernel-time share)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import alpine
shapes = [(96, 80), (128, 112), (160, 128), (192, 160)]
G = len(shapes)
gl = [alpine(nx, ny) for nx, ny in shapes]
b = odinn.GlacierBatch(shapes, [50.0] * G)
for k, (H0, B) in enumerate(gl):
    b.set_fields(k, H0, B)
ph = odinn.PhysicalParameters()
nn = odinn.NeuralNetwork(odinn.Parameters(), seed=666)
mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA)
b.set_law(odinn.LAW_NN_A_SCALAR, mlp, nn.theta)
ts = [2010.0 + k / 12.0 for k in range(25)]
for k in range(G):
    b.set_reference(k, ts, [gl[k][0] * (1.0 - 0.002 * j) for j in range(len(ts))], 3)
def tm(f, n=20):
    f(); b.sync()
    t0 = time.perf_counter()
    for _ in range(n): f()
    b.sync()
    return (time.perf_counter() - t0) / n * 1e3
st = b.solve(ts, reltol=1e-8)
print("steps", [(s.naccept, s.nreject) for s in st])
print("solve ms", tm(lambda: b.solve(ts, reltol=1e-8)))
print("loss_grad ms", tm(lambda: b.loss_grad(ts, theta=nn.theta, reltol=1e-8)))
print("continuous ms", tm(lambda: b.loss_grad_continuous(ts, theta=nn.theta, reltol=1e-8), n=5), b.last_stats_rev[0].naccept, b.last_stats_rev[0].nreject)
