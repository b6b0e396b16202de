"""LossHV gradient evaluations at 8 x 1024^2 (k = 13 monthly snapshots with thickness AND velocity data, scalar NN law,
reltol 1e-8): wall time of both adjoints (run under rocprofv3 --kernel-trace --stats for the kernel shares).
usage: velocity_probe.py [n] [G] [law: A (scalar NN law, default) | U (target :D, default 2-3-10-3-1 net, f = 0.8)]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lawname = sys.argv[3] if len(sys.argv) > 3 else "A"
gl = [make_glacier(n, k) for k in range(G)]
b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl])
for k, (H0, B, A) in enumerate(gl):
    b.set_fields(k, H0, B)
ph = odinn.PhysicalParameters()
nn = odinn.NeuralNetwork(odinn.Parameters(), seed=666)
mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA)
theta = nn.theta
b.set_law(odinn.LAW_NN_A_SCALAR, mlp, nn.theta)
if lawname == "U":
    mlp = odinn.MLPSpec([2, 3, 10, 3, 1], [odinn.ACT_SOFTPLUS] * 3 + [odinn.ACT_SIGMOID], [(0.0, 300.0), (0.0, 0.5)], odinn.POST_EXPMAX, 0.0, 50.0)
    theta = np.random.default_rng(1234).uniform(-0.5, 0.5, mlp.n_params)
    b.set_law(odinn.LAW_NN_U, mlp, theta)
    b.set_surface_velocity_factor(0.8)
ts = [2010.0 + k / 12.0 for k in range(13)]
b.solve(ts, reltol=1e-8)
for k in range(G):
    H0 = gl[k][0]
    b.set_reference(k, ts, [H0 * (1.0 - 0.002 * j) for j in range(len(ts))], 3)
    Vx, Vy = b.surface_V(k, H0)
    Va = np.hypot(Vx, Vy)
    b.set_velocity_reference(k, ts, [0.9 * Va] * len(ts), [0.9 * Vx] * len(ts), [0.9 * Vy] * len(ts))
def tm(f, n=3):
    f(); b.sync()
    t0 = time.perf_counter()
    for _ in range(n): f()
    b.sync()
    return (time.perf_counter() - t0) / n * 1e3
for kind, name in ((odinn._lib.LOSS_H, "LossH"), (odinn._lib.LOSS_HV, "LossHV"), (odinn._lib.LOSS_V, "LossV")):
    b.set_loss(kind, "xy", True, 1.0)
    print(name, "discrete ms %.2f" % tm(lambda: b.loss_grad(ts, theta=theta, reltol=1e-8)),
          "continuous ms %.2f" % tm(lambda: b.loss_grad_continuous(ts, theta=theta, reltol=1e-8), n=1),
          b.last_stats_rev[0].naccept, b.last_stats_rev[0].nreject)
