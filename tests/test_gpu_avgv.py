"""GPU: LossAvgV, the time-averaged surface-velocity loss (src/losses/TimeAggregatedLosses.jl:115-258) -- the loss term,
dL/dH of every point of its time grid and dL/dtheta inside odinn_loss / odinn_loss_grad (DiscreteAdjoint,
gradient.jl:170-215,274) / odinn_loss_grad_continuous (ContinuousAdjoint, :369-449,538) against the oracle's restatement:
alone and next to LossH (MultiLoss weights), :xy and :abs, scalar and gridded A laws, a mass balance, ragged batches with
different windows per glacier; through the reference-facing API as well."""
import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays
from oracle import sia2d_oracle as O
from test_gpu_parity import _inversion_case

pytestmark = pytest.mark.gpu


def _sample(gl, law, cfg, ts, i1, i2, component, scale=1.15):
    """One velocity sample over [ts[i1], ts[i2]]: the time average of a run, scaled."""
    snaps, _, _ = O.forward(gl, law, cfg)
    a = O.AvgVData(ts[i1], ts[i2], None, None, None, component, ts[1] - ts[0])
    tl, dt = O.avgv_times(a)
    assert len(tl) == i2 - i1
    V = [O.V_from_H(snaps[i1 + i], gl.B, gl.dx, gl.dy, gl.phys, law) for i in range(len(tl))]
    a.Vx = scale * sum(v[0] * d for v, d in zip(V, dt)) / sum(dt)
    a.Vy = scale * sum(v[1] * d for v, d in zip(V, dt)) / sum(dt)
    a.Vabs = np.sqrt(a.Vx ** 2 + a.Vy ** 2)
    return a


@pytest.mark.parametrize("case", ["with_H_xy", "alone_xy", "alone_abs", "with_H_mb", "whole_span"])
def test_avgv_loss_and_gradients_match_oracle(gpu, case):
    nx, ny = 64, 48
    use_mb = case == "with_H_mb"
    comp = "abs" if case.endswith("abs") else "xy"
    ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref = _inversion_case(gpu, nx, ny, use_mb)
    i1, i2 = (0, len(ts) - 1) if case == "whole_span" else (1, len(ts) - 2)
    law0 = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th0, T=-2.0)
    a = _sample(gl, law0, cfg, ts, i1, i2, comp)
    cfg.avgv, cfg.avgv_weight = a, 2.0
    alone = case.startswith("alone")
    Href, tH = ([], []) if alone else (ref, ts)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-2.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_A_SCALAR, gm, th0)
    if Href:
        b.set_reference(0, ts, ref, 3)
    if mb is not None:
        b.set_mass_balance(0, mb.mb0, mb.dmb_dS, mb.S_ref, mb.mb_max)
    b.set_avgv_reference(0, a.t1, a.t2, a.Vabs, a.Vx, a.Vy)
    b.set_avgv_loss(2.0, a.step, comp)
    mbt = ts[1:] if use_mb else ()
    # forward loss (inversion_utils.jl:457-460)
    b.solve(ts, mb_times=mbt, reltol=1e-8)
    snaps, _, _ = O.forward(gl, law0, cfg)
    lo_fwd = (O.loss_H(snaps, ts, Href, tH, 3) if Href else 0.0) + O.avgv_loss_terms(snaps, ts, cfg, gl, law0)[0]
    assert abs(b.loss()[0] - lo_fwd) <= 1e-6 * abs(lo_fwd)
    # discrete adjoint
    Lo, go, lam0 = O.loss_and_grad(gl, law0, cfg, Href, tH)
    Lg, gg = b.loss_grad(ts, theta=th0, mb_times=mbt, reltol=1e-8)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (case, ratio, angle, relerr)
    # lambda(t0) of the DiscreteAdjoint is NOT compared on adaptive solves: its reverse loop is an explicit Euler
    # recursion with the snapshot spacing as step (gradient.jl:242), which amplifies the grid-scale content of the
    # velocity cotangents -- in the oracle alone, reltol 1e-8 -> 1e-10 moves the snapshots by 2.5e-8, dL/dtheta by 1.4e-7
    # and lambda(t0) by 1.4e-3 (row-alternating pattern); the theta-VJP does not see that component.  The fixed-step
    # test below compares lambda(t0) where both sides follow the same arithmetic sequence.
    # continuous adjoint
    adj = O.ContinuousAdjointCfg(n_quadrature=24)
    Lo, go, lam0, st_o = O.loss_and_grad_continuous(gl, law0, cfg, Href, tH, adj)
    Lg, gg = b.loss_grad_continuous(ts, theta=th0, mb_times=mbt, reltol=1e-8, n_quadrature=24)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (case, ratio, angle, relerr)
    assert rel_l2(b.lambda0(0), lam0) < 1e-4
    # the term really is in there
    b.set_avgv_loss(0.0, a.step, comp)
    if Href:
        L2, g2 = b.loss_grad(ts, theta=th0, mb_times=mbt, reltol=1e-8)
        assert abs(L2 - Lg) > 1e-3 * abs(Lg) and rel_l2(g2, gg) > 1e-3
    else:
        with pytest.raises(gpu.OdinnError):
            b.loss_grad(ts, theta=th0, reltol=1e-8)
    b.close()


def test_avgv_lambda0_fixed_step(gpu):
    """lambda(t0) of both adjoints with LossAvgV alone on fixed-step forward solves (same arithmetic sequence on both
    sides, snapshots equal to rounding): dL/dH of every stop of the time grid, propagated through the reverse loops."""
    nx, ny = 64, 48
    ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref = _inversion_case(gpu, nx, ny, False)
    law0 = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th0, T=-2.0)
    fdt = (ts[1] - ts[0]) / 16.0
    cfg.fixed_dt = fdt
    a = _sample(gl, law0, cfg, ts, 0, len(ts) - 1, "xy")
    cfg.avgv, cfg.avgv_weight = a, 1.0
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-2.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_A_SCALAR, gm, th0)
    b.set_avgv_reference(0, a.t1, a.t2, a.Vabs, a.Vx, a.Vy)
    b.set_avgv_loss(1.0, a.step, "xy")
    Lo, go, lam0 = O.loss_and_grad(gl, law0, cfg, [], [])
    Lg, gg = b.loss_grad(ts, theta=th0, fixed_dt=fdt)
    assert abs(Lg - Lo) <= 1e-10 * abs(Lo) and rel_l2(gg, go) < 1e-9
    assert rel_l2(b.lambda0(0), lam0) < 1e-6
    Lo, go, lam0, _ = O.loss_and_grad_continuous(gl, law0, cfg, [], [], O.ContinuousAdjointCfg(n_quadrature=24))
    Lg, gg = b.loss_grad_continuous(ts, theta=th0, fixed_dt=fdt, n_quadrature=24)
    assert abs(Lg - Lo) <= 1e-10 * abs(Lo) and rel_l2(gg, go) < 1e-6
    assert rel_l2(b.lambda0(0), lam0) < 1e-5
    b.close()


def test_avgv_ragged_batch_gridded_law_and_errors(gpu):
    """Three ragged glaciers, a hoisted gridded A = NN(T) law (theta-part through the dual-grid accumulator), different
    windows per glacier and one glacier without a sample; batch result = sum of the per-glacier oracle results.  Times
    off the stops and Y/U laws are rejected."""
    ph = O.Phys()
    from test_gpu_parity import _mlp_pair
    om, gm, th = _mlp_pair(gpu, [1, 3, 10, 3, 1], [1, 1, 1, 2], None, O.POST_AFFINE, ph.minA, ph.maxA)
    shapes = [(56, 40), (80, 48), (40, 33)]
    step = 1.0 / 48.0
    ts = [2010.0 + j * step for j in range(7)]
    windows = [(0, 6), (2, 5), None]
    rng = np.random.default_rng(4)
    b = gpu.GlacierBatch(shapes, [50.0] * 3)
    gls, laws, cfgs, refs = [], [], [], []
    for k, (nx, ny) in enumerate(shapes):
        H0, B = O.synthetic_alpine(nx, ny, hmax=150.0, slope=0.1)
        T = np.asfortranarray(-5.0 - 4.0 * rng.uniform(size=(nx - 1, ny - 1)))
        b.set_fields(k, H0, B)
        b.set_T_field(k, T)
        gl = O.Glacier(H0, B, 50.0, 50.0, ph)
        law = O.Law(kind=O.LAW_NN_A_GRIDDED, mlp=om, theta=th, T=T)
        cfg = O.SimConfig(tstops=ts, reltol=1e-8)
        ref, _, _ = O.forward(gl, law, cfg)
        ref = [r * (1.0 + 0.02 * j) for j, r in enumerate(ref)]
        b.set_reference(k, ts, ref, 3)
        if windows[k] is not None:
            a = _sample(gl, law, cfg, ts, windows[k][0], windows[k][1], "xy")
            cfg.avgv, cfg.avgv_weight = a, 0.5
            b.set_avgv_reference(k, a.t1, a.t2, a.Vabs, a.Vx, a.Vy)
        gls.append(gl); laws.append(law); cfgs.append(cfg); refs.append(ref)
    b.set_law(gpu.LAW_NN_A_GRIDDED, gm, th)
    b.set_avgv_loss(0.5, step, "xy")
    Lo, go = 0.0, 0.0
    Lc, gc = 0.0, 0.0
    for k in range(3):
        l, g, _ = O.loss_and_grad(gls[k], laws[k], cfgs[k], refs[k], ts)
        Lo, go = Lo + l, go + g
        l, g, _, _ = O.loss_and_grad_continuous(gls[k], laws[k], cfgs[k], refs[k], ts, O.ContinuousAdjointCfg(n_quadrature=8))
        Lc, gc = Lc + l, gc + g
    Lg, gg = b.loss_grad(ts, theta=th, reltol=1e-8)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo) and rel_l2(gg, go) < 1e-5
    Lg, gg = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
    assert abs(Lg - Lc) <= 1e-6 * abs(Lc) and rel_l2(gg, gc) < 1e-5
    # a window whose grid leaves the stops
    a = cfgs[0].avgv
    b.set_avgv_reference(0, a.t1 + 1e-3, a.t2, a.Vabs, a.Vx, a.Vy)
    with pytest.raises(gpu.OdinnError, match="not among the tstops"):
        b.loss_grad(ts, theta=th, reltol=1e-8)
    b.close()


def test_multiloss_with_lossavgv_through_the_api(gpu):
    """MultiLoss((LossH(), LossAvgV(step)), (1.5, 0.7)) on two ragged glaciers with different velocity windows through
    Inversion / SIA2D_grad_b (per-glacier classical law) against the oracle; the stops gain the points of the time grids."""
    k, step = 7, 1.0 / 96.0
    p = gpu.Parameters(simulation=gpu.SimulationParameters(tspan=(2010.0, 2010.0 + (k - 1) * step)),
                       solver=gpu.SolverParameters(reltol=1e-10, step=2 * step),
                       hyper=gpu.Hyperparameters(optimizer=gpu.LBFGS(), epochs=3))
    p.UDE.grad = gpu.DiscreteAdjoint()
    p.UDE.empirical_loss_function = gpu.MultiLoss(losses=(gpu.LossH(), gpu.LossAvgV(step=step)), lambdas=(1.5, 0.7))
    ts = [2010.0 + j * step for j in range(k)]
    tH = ts[::2]
    ph = O.Phys()
    gl, samples = [], []
    for kk, (nx, ny) in enumerate([(48, 40), (64, 48)]):
        H0, B = O.synthetic_alpine(nx, ny, hmax=160.0, slope=0.1)
        g = gpu.Glacier2D(f"SYN-{kk}", H0, B, 50.0, 50.0, A=3e-17)
        g.thicknessData = gpu.ThicknessData(tH, [H0 * (1.0 - 0.01 * j) for j in range(len(tH))])
        og = O.Glacier(H0, B, 50.0, 50.0, ph)
        a = _sample(og, O.Law(kind=O.LAW_CONST_A, A=5e-17), O.SimConfig(tstops=ts, reltol=1e-10), ts, kk, 6, "xy")
        g.velocityData = gpu.VelocityData(t=[], vabs=[a.Vabs], vx=[a.Vx], vy=[a.Vy], date1=[a.t1], date2=[a.t2])
        gl.append(g); samples.append(a)
    reg = gpu.GlacierWideInv(p, gl, "A")
    inv = gpu.Inversion(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.LawA(p, scalar=True)), regressors={"A": reg}), gl, p)
    assert np.allclose(inv.tstops(), ts, rtol=0, atol=1e-12) and len(inv.tstops()) == k
    th = reg.theta.copy()
    dth = np.zeros_like(th)
    L = gpu.SIA2D_grad_b(dth, th, inv)
    lo, hi = ph.minA, ph.maxA
    Lo, go = 0.0, np.zeros(2)
    for kk, g in enumerate(gl):
        A = lo + (hi - lo) * (np.tanh(th[kk]) + 1) / 2
        # (every glacier has its OWN stop table: the shared grid, its data times, the grid of its velocity window)
        cfg = O.SimConfig(tstops=inv.tstops_glacier(kk), reltol=1e-10, avgv=samples[kk], avgv_weight=0.7 / 1.5)
        l1, g1, _ = O.loss_and_grad(O.Glacier(g.H0, g.B, 50.0, 50.0, ph), O.Law(kind=O.LAW_CONST_A, A=A), cfg, g.thicknessData.H, tH)
        Lo += 1.5 * l1
        go[kk] = 1.5 * g1[0] * (hi - lo) / 2 * (1 - np.tanh(th[kk]) ** 2)
    assert abs(L - Lo) <= 1e-6 * abs(Lo)
    assert np.allclose(dth, go, rtol=1e-5)
    with pytest.raises(ValueError, match="exactly one sample"):
        gl[0].velocityData = gpu.VelocityData(t=[], vabs=[], vx=[], vy=[])
        gpu.Inversion(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.LawA(p, scalar=True)), regressors={"A": gpu.GlacierWideInv(p, gl, "A")}), gl, p).tstops()
