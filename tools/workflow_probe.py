"""Gradient evaluations of other workflows at scale (run under rocprofv3 --kernel-trace --stats for the kernel shares):
  python tools/workflow_probe.py gridded [n G]    A = NN(T) on the dual grid (dual-grid accumulator), LossH and LossHV
  python tools/workflow_probe.py mb [n G]         scalar NN law + linear mass balance at every stop, LossH
  python tools/workflow_probe.py Y [n G]          Y = NN(T, Hbar) (target :D_hybrid, default :Linear interpolation), LossH
  python tools/workflow_probe.py U [n G]          U = NN(Hbar, |grad S|) (target :D), LossH
  python tools/workflow_probe.py U n G scaled      LawU with prescale_bounds = [(0, 300), (0, 0.5)], max_NN = 50 (the table can follow it)
  python tools/workflow_probe.py Y|U n G custom   the same with a run-time architecture (2-5-10-5-1, gelu x3 + softplus: law mode 2)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier, temperature_field
what = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
k = 13
gl = [make_glacier(n, j) for j in range(G)]
b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl], T=[-5.0] * G)
for j, (H0, B, A) in enumerate(gl):
    b.set_fields(j, H0, B)
ph = odinn.PhysicalParameters()
P = odinn.Parameters()
rng = np.random.default_rng(1234)
if what == "gridded":
    for j, (H0, B, A) in enumerate(gl):
        b.set_T_field(j, temperature_field(H0, B))
    mlp = odinn.MLPSpec([1, 16, 16, 1], [odinn.ACT_SOFTPLUS, odinn.ACT_SOFTPLUS, odinn.ACT_SIGMOID], None, odinn.POST_AFFINE, ph.minA, ph.maxA)
    theta = rng.uniform(-0.5, 0.5, mlp.n_params)
    b.set_law(odinn.LAW_NN_A_GRIDDED, mlp, theta)
elif what == "mb":
    nn = odinn.NeuralNetwork(P, seed=666)
    mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA)
    theta = nn.theta
    b.set_law(odinn.LAW_NN_A_SCALAR, mlp, theta)
    for j, (H0, B, A) in enumerate(gl):
        S = B + H0
        b.set_mass_balance(j, np.full_like(H0, -0.05), 1e-4, np.full_like(H0, float(S.mean())), 0.2)
else:
    kwU = {}
    if what == "U" and len(sys.argv) > 4 and sys.argv[4] == "scaled":  # LawU as the reference's tests build it (test/SIA2D_adjoint.jl): inputs scaled, U <= 50 m/yr
        kwU = dict(prescale_bounds=[(0.0, 300.0), (0.0, 0.5)], max_NN=50.0)
    model = odinn.SIA2Dmodel(P, **{what: (odinn.LawY if what == "Y" else odinn.LawU)(odinn.NeuralNetwork(P, architecture=odinn.build_default_NN(2), seed=666), P, **kwU)})
    law = model.law
    theta = law.nn.theta
    mlp = law.mlp
    if len(sys.argv) > 4 and sys.argv[4] == "custom":
        mlp = odinn.MLPSpec([2, 5, 10, 5, 1], [odinn.ACT_GELU] * 3 + [odinn.ACT_SOFTPLUS], law.mlp.prescale, law.mlp.post_kind, law.mlp.post_lo, law.mlp.post_hi)
        theta = rng.uniform(-0.5, 0.5, mlp.n_params)
    elif len(sys.argv) > 4 and sys.argv[4] == "custom_sp":  # the default activations on other widths: isolates the run-time loops
        mlp = odinn.MLPSpec([2, 4, 10, 3, 1], [odinn.ACT_SOFTPLUS] * 3 + [odinn.ACT_SIGMOID], law.mlp.prescale, law.mlp.post_kind, law.mlp.post_lo, law.mlp.post_hi)
        theta = rng.uniform(-0.5, 0.5, mlp.n_params)
    b.set_law(law.kind, mlp, theta, law.n_H, law.n_gradS)
ts = [2010.0 + j / 12.0 for j in range(k)]
mbt = ts[1:] if what == "mb" else ()
b.solve(ts, mb_times=mbt, reltol=1e-8)
for j in range(G):
    H0 = gl[j][0]
    b.set_reference(j, ts, [H0 * (1.0 - 0.002 * i) for i in range(k)], 3)
def tm(f, nrep=2):
    f(); b.sync()
    t0 = time.perf_counter()
    for _ in range(nrep): f()
    b.sync()
    return (time.perf_counter() - t0) / nrep * 1e3
print(what, n, G, "solve ms %.2f" % tm(lambda: b.solve(ts, mb_times=mbt, reltol=1e-8)), [(s.naccept, s.nreject) for s in b.solve(ts, mb_times=mbt, reltol=1e-8)][:1])
print("LossH discrete ms %.2f" % tm(lambda: b.loss_grad(ts, theta=theta, mb_times=mbt, reltol=1e-8)))
print("LossH continuous ms %.2f" % tm(lambda: b.loss_grad_continuous(ts, theta=theta, mb_times=mbt, reltol=1e-8), 1), b.last_stats_rev[0].naccept, b.last_stats_rev[0].nreject)
if what == "gridded":
    for j in range(G):
        Vx, Vy = b.surface_V(j, gl[j][0])
        b.set_velocity_reference(j, ts, [0.9 * np.hypot(Vx, Vy)] * k, [0.9 * Vx] * k, [0.9 * Vy] * k)
    b.set_loss(odinn._lib.LOSS_HV, "xy", True, 1.0)
    print("LossHV discrete ms %.2f" % tm(lambda: b.loss_grad(ts, theta=theta, reltol=1e-8)))
    print("LossHV continuous ms %.2f" % tm(lambda: b.loss_grad_continuous(ts, theta=theta, reltol=1e-8), 1), b.last_stats_rev[0].naccept)
