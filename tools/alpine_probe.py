"""512 alpine glaciers (configs[3]-like, cycling 96x80 ... 192x160), k = 25, reltol 1e-8: wall time of the solve and of both
gradients (run under rocprofv3 --kernel-trace --stats for the kernel shares)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import alpine
base = [(96, 80), (128, 112), (160, 128), (192, 160)]
G = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shapes = [base[k % 4] for k in range(G)]
gl = [alpine(nx, ny) for nx, ny in base]
b = odinn.GlacierBatch(shapes, [50.0] * G)
for k in range(G):
    b.set_fields(k, *gl[k % 4])
ph = odinn.PhysicalParameters()
nn = odinn.NeuralNetwork(odinn.Parameters(), seed=666)
mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA)
b.set_law(odinn.LAW_NN_A_SCALAR, mlp, nn.theta)
ts = [2010.0 + k / 12.0 for k in range(25)]
for k in range(G):
    b.set_reference(k, ts, [gl[k % 4][0] * (1.0 - 0.002 * j) for j in range(len(ts))], 3)
def tm(f, n=5):
    f(); b.sync()
    t0 = time.perf_counter()
    for _ in range(n): f()
    b.sync()
    return (time.perf_counter() - t0) / n * 1e3
print("solve ms", tm(lambda: b.solve(ts, reltol=1e-8)))
print("loss_grad ms", tm(lambda: b.loss_grad(ts, theta=nn.theta, reltol=1e-8)))
print("continuous ms", tm(lambda: b.loss_grad_continuous(ts, theta=nn.theta, reltol=1e-8), n=2), b.last_stats_rev[0].naccept, b.last_stats_rev[0].nreject)
