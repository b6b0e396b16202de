"""Time-aggregated / regularisation losses at scale (8 x 1024^2, scalar NN law, k = 13): LossH + one extra term each.
Run under rocprofv3 --kernel-trace --stats for the kernel shares."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k = 13
gl = [make_glacier(n, j) for j in range(G)]
ph = odinn.PhysicalParameters()
nn = odinn.NeuralNetwork(odinn.Parameters(), seed=666)
mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA)
ts = [2010.0 + j / 12.0 for j in range(k)]
def tm(b, f, nrep=2):
    f(); b.sync()
    t0 = time.perf_counter()
    for _ in range(nrep): f()
    b.sync()
    return (time.perf_counter() - t0) / nrep * 1e3
for what in os.environ.get("AGG_WHAT", "plain,dhdt,avgv,vreg").split(","):
    b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl])
    for j, (H0, B, A) in enumerate(gl):
        b.set_fields(j, H0, B)
    b.set_law(odinn.LAW_NN_A_SCALAR, mlp, nn.theta)
    for j in range(G):
        H0 = gl[j][0]
        b.set_reference(j, ts, [H0 * (1.0 - 0.002 * i) for i in range(k)], 3)
        if what == "dhdt":
            b.set_dhdt_reference(j, ts[0], ts[-1], -1.0)
        if what in ("avgv", "vreg"):
            Vx, Vy = b.surface_V(j, H0)
            Va = np.hypot(Vx, Vy)
            if what == "avgv":
                b.set_avgv_reference(j, ts[0], ts[-1], 0.9 * Va, 0.9 * Vx, 0.9 * Vy)
            else:
                b.set_velocity_reference(j, ts, [0.9 * Va] * k, [0.9 * Vx] * k, [0.9 * Vy] * k)
    if what == "dhdt": b.set_dhdt_loss(1.0)
    if what == "avgv": b.set_avgv_loss(1.0, 1.0 / 12.0, "xy")
    if what == "vreg": b.set_velocity_regularization(1.0, int(os.environ.get("VREG_DIST", "3")))
    print(what, "discrete ms %.2f" % tm(b, lambda: b.loss_grad(ts, theta=nn.theta, reltol=1e-8)),
          "continuous ms %.2f" % tm(b, lambda: b.loss_grad_continuous(ts, theta=nn.theta, reltol=1e-8), 1), flush=True)
    b.close()
