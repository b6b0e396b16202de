"""Device-memory growth over repeated gradient evaluations of several workflows (hipMemGetInfo via torch): must be flat."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import _odinn_import
odinn = _odinn_import.load()
from bench import alpine
shapes = [(96, 80), (128, 112), (160, 128), (192, 160)] * 4
G = len(shapes)
ph = odinn.PhysicalParameters(); P = odinn.Parameters()
ts = [2010.0 + k / 12.0 for k in range(7)]
def used(): f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
for what in ("scalar", "gridded", "Y", "U"):
    b = odinn.GlacierBatch(shapes, [50.0] * G, T=[-5.0] * G)
    fields = [alpine(*s) for s in shapes[:4]]
    for k in range(G):
        b.set_fields(k, *fields[k % 4])
        b.set_reference(k, ts, [fields[k % 4][0] * (1 - 0.01 * j) for j in range(7)], 3)
    if what == "scalar":
        nn = odinn.NeuralNetwork(P, seed=1); mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA); th = nn.theta
        b.set_law(odinn.LAW_NN_A_SCALAR, mlp, th)
    elif what == "gridded":
        nn = odinn.NeuralNetwork(P, seed=1); mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA); th = nn.theta
        for k, (nx, ny) in enumerate(shapes): b.set_T_field(k, np.full((nx - 1, ny - 1), -5.0))
        b.set_law(odinn.LAW_NN_A_GRIDDED, mlp, th)
    else:
        law = (odinn.LawY if what == "Y" else odinn.LawU)(odinn.NeuralNetwork(P, architecture=odinn.build_default_NN(2), seed=1), P)
        th = law.nn.theta
        b.set_law(law.kind, law.mlp, th, law.n_H, law.n_gradS)
    b.loss_grad(ts, theta=th, reltol=1e-6); b.loss_grad_continuous(ts, theta=th, reltol=1e-6, n_quadrature=8); b.sync()
    m0 = used()
    for _ in range(15):
        b.loss_grad(ts, theta=th, reltol=1e-6); b.loss_grad_continuous(ts, theta=th, reltol=1e-6, n_quadrature=8)
    b.sync()
    print(what, "MiB used before/after 15 more evaluation pairs: %.1f %.1f" % (m0, used()), flush=True)
    b.close()
print("after closing all batches: %.1f MiB" % used())
