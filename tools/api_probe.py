"""Host-layer overhead: SIA2D_grad_b through the API (Inversion / Model / LawA) against the bare GlacierBatch call on the same
batch (8 x 512^2, scalar NN law, LossH, k = 13): the API must add microseconds, not uploads."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier
n, G, k = 512, 8, 13
step = 1.0 / 12.0
for adj in ("DiscreteAdjoint", "ContinuousAdjoint"):
    p = odinn.Parameters(simulation=odinn.SimulationParameters(tspan=(2010.0, 2010.0 + (k - 1) * step)),
                         solver=odinn.SolverParameters(reltol=1e-8, step=step))
    p.UDE.grad = getattr(odinn, adj)()
    gls = []
    ts = [2010.0 + j * step for j in range(k)]
    for j in range(G):
        H0, B, A = make_glacier(n, j)
        g = odinn.Glacier2D(rgi_id="g%d" % j, H0=H0, B=B, dx=100.0, dy=100.0)
        g.thicknessData = odinn.ThicknessData(ts, [H0 * (1 - 0.002 * i) for i in range(k)])
        gls.append(g)
    nn = odinn.NeuralNetwork(p, seed=1)
    inv = odinn.FunctionalInversion(odinn.Model(odinn.SIA2Dmodel(p, A=odinn.LawA(nn, p)), regressors={"A": nn}), gls, p)
    th = nn.theta.copy(); dth = np.zeros_like(th)
    odinn.SIA2D_grad_b(dth, th, inv); odinn.SIA2D_grad_b(dth, th, inv)
    t0 = time.perf_counter()
    for _ in range(3): odinn.SIA2D_grad_b(dth, th, inv)
    t_api = (time.perf_counter() - t0) / 3 * 1e3
    b = inv.batch()
    f = (lambda: b.loss_grad(inv.tstops(), theta=th, reltol=1e-8)) if adj == "DiscreteAdjoint" else \
        (lambda: b.loss_grad_continuous(inv.tstops(), theta=th, reltol=1e-8))
    f()
    t0 = time.perf_counter()
    for _ in range(3): f()
    b.sync()
    t_raw = (time.perf_counter() - t0) / 3 * 1e3
    print(adj, "API ms %.2f  bare batch ms %.2f" % (t_api, t_raw), flush=True)
