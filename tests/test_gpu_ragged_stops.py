"""GPU: per-glacier stop tables.  The reference builds tstops PER GLACIER -- the `step` grid and solver.tstops are shared,
the thickness / velocity data times are the glacier's own (src/simulations/inversions/inversion_utils.jl:487-495,
src/inverse/SIA2D/gradient.jl:96-107): a glacier's integrator never lands on another glacier's data times and both reverse
loops walk the glacier's own stops only.  And the mass-balance PeriodicCallback (:498-517) inserts integrator stops of its
own without adding snapshots to the result (step_MB not a multiple of solver.step).

Checked here: a batch of glaciers with different data times == every glacier solved alone == the oracle's per-glacier
restatement, for the forward solve and both adjoints; a Prediction with step_MB = 1/24, solver.step = 1/12 against the oracle."""
import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays, sched_env
from oracle import sia2d_oracle as O
from test_gpu_parity import _mb

pytestmark = pytest.mark.gpu

T0 = 2010.0


def _two_glaciers(gpu, with_mb=False):
    """Two valley glaciers of different size whose thickness data sit at different times (only t0 and t1 shared)."""
    ph = O.Phys()
    shapes = [(64, 48), (40, 56)]
    t1 = T0 + 6.0 / 240.0
    own = [[T0, T0 + 2.0 / 240.0, T0 + 4.0 / 240.0, t1],
           [T0, T0 + 1.0 / 240.0, T0 + 3.0 / 240.0, T0 + 4.5 / 240.0, t1]]
    om = O.default_nn(1, light=False, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    gm = gpu.MLPSpec(om.widths, om.acts, None, O.POST_AFFINE, ph.minA, ph.maxA)
    th_true = om.init_theta(np.random.default_rng(42))
    th0 = om.init_theta(np.random.default_rng(1234))
    gls, refs, mbs, cfgs = [], [], [], []
    for (nx, ny), ts in zip(shapes, own):
        H0, B = O.synthetic_valley(nx, ny, 50.0)
        gl = O.Glacier(H0, B, 50.0, 50.0, ph)
        mb = _mb(H0, B, step=2.0 / 240.0) if with_mb else None
        gls.append(gl)
        mbs.append(mb)
    return ph, shapes, own, om, gm, th_true, th0, gls, mbs


def _make_batch(gpu, idx, shapes, gls, gm, th0, own, refs, mbs, T=-2.0):
    b = gpu.GlacierBatch([shapes[i] for i in idx], [50.0] * len(idx), T=[T] * len(idx))
    for k, i in enumerate(idx):
        b.set_fields(k, gls[i].H0, gls[i].B)
        b.set_reference(k, own[i], refs[i], 3)
        if mbs[i] is not None:
            b.set_mass_balance(k, mbs[i].mb0, mbs[i].dmb_dS, mbs[i].S_ref, mbs[i].mb_max)
    b.set_law(gpu.LAW_NN_A_SCALAR, gm, th0)
    return b


@pytest.mark.parametrize("with_mb", [False, True])
def test_ragged_data_times_forward_and_discrete_adjoint(gpu, with_mb):
    """Fixed dt (clipped at each glacier's OWN stops): the batch == each glacier alone == the oracle to 1e-10."""
    ph, shapes, own, om, gm, th_true, th0, gls, mbs = _two_glaciers(gpu, with_mb)
    dt = 1.0 / 960.0
    union = sorted(set(own[0]) | set(own[1]))
    # (DiscreteAdjoint: the MB times must be stops of every glacier, gradient.jl:131 -- t1 is)
    mbt = [own[0][-1]] if with_mb else []
    refs, Lo, go, lam0 = [], [], [], []
    for i in range(2):
        cfg_t = O.SimConfig(tstops=own[i], fixed_dt=dt, mb=mbs[i], mb_times=mbt)
        ref, _, _ = O.forward(gls[i], O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th_true, T=-2.0), cfg_t)
        refs.append(ref)
        l, g, l0 = O.loss_and_grad(gls[i], O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th0, T=-2.0), cfg_t, ref, own[i])
        Lo.append(l); go.append(g); lam0.append(l0)
    # --- the batch with per-glacier tables
    b = _make_batch(gpu, [0, 1], shapes, gls, gm, th0, own, refs, mbs)
    for k in range(2):
        b.set_glacier_stops(k, own[k])
    Lb, gb = b.loss_grad(union, theta=th0, mb_times=mbt, fixed_dt=dt)
    lossg, _ = b.grad_parts()
    snaps_b = [[b.snapshot(k, j) for j in range(len(own[k]))] for k in range(2)]
    lam_b = [b.lambda0(k) for k in range(2)]
    with pytest.raises(Exception):
        b.snapshot(0, len(own[0]))  # glacier 0 has 4 stops of its own, not the 5 of glacier 1
    b.close()
    assert abs(Lb - sum(Lo)) <= 1e-10 * abs(sum(Lo))
    assert rel_l2(gb, go[0] + go[1]) < 1e-10
    for k in range(2):
        assert abs(lossg[k] - Lo[k]) <= 1e-10 * abs(Lo[k])
        assert rel_l2(lam_b[k], lam0[k]) < 1e-10
        fo, _, _ = O.forward(gls[k], O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th0, T=-2.0),
                             O.SimConfig(tstops=own[k], fixed_dt=dt, mb=mbs[k], mb_times=mbt))
        for j in range(len(own[k])):
            assert rel_l2(snaps_b[k][j], fo[j]) < 1e-11, (k, j)
    # --- every glacier alone (its own table is THE table of the call)
    for i in range(2):
        b1 = _make_batch(gpu, [i], shapes, gls, gm, th0, own, refs, mbs)
        L1, g1 = b1.loss_grad(own[i], theta=th0, mb_times=mbt, fixed_dt=dt)
        assert abs(L1 - Lo[i]) <= 1e-10 * abs(Lo[i])
        assert rel_l2(g1, go[i]) < 1e-10
        assert rel_l2(b1.lambda0(0), lam_b[i]) < 1e-11
        b1.close()
    # --- and the union table for everybody is NOT the reference's result (extra reverse-Euler points)
    b = _make_batch(gpu, [0, 1], shapes, gls, gm, th0, own, refs, mbs)
    Lu, gu = b.loss_grad(union, theta=th0, mb_times=mbt, fixed_dt=dt)
    b.close()
    assert rel_l2(gu, go[0] + go[1]) > 1e-6


def test_ragged_data_times_adaptive_solve_lands_on_own_stops_only(gpu):
    ph, shapes, own, om, gm, th_true, th0, gls, mbs = _two_glaciers(gpu)
    union = sorted(set(own[0]) | set(own[1]))
    refs = [[gls[i].H0] * len(own[i]) for i in range(2)]
    b = _make_batch(gpu, [0, 1], shapes, gls, gm, th0, own, refs, mbs)
    for k in range(2):
        b.set_glacier_stops(k, own[k])
    st = b.solve(union, reltol=1e-8)
    snaps = [[b.snapshot(k, j) for j in range(len(own[k]))] for k in range(2)]
    b.set_glacier_stops(0, None)  # cleared: glacier 0 takes the union again
    b.set_glacier_stops(1, None)
    st_u = b.solve(union, reltol=1e-8)
    b.close()
    law = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th0, T=-2.0)
    for k in range(2):
        fo, so, _ = O.forward(gls[k], law, O.SimConfig(tstops=own[k], reltol=1e-8))
        for j in range(len(own[k])):
            assert rel_l2(snaps[k][j], fo[j]) < 1e-6, (k, j)
        # the same controller on the same stops: same step counts (a step of slack for an estimate on the accept edge)
        assert abs(st[k].naccept - so.naccept) <= 2 and abs(st[k].nreject - so.nreject) <= 2, (k, st[k], so)
        assert abs(st[k].t_final - own[k][-1]) < 1e-12
    # with the union table every glacier also stops on the other's data times: another step sequence
    assert (st_u[0].naccept, st_u[1].naccept) != (st[0].naccept, st[1].naccept)


@pytest.mark.parametrize("fused", ["1", "0"])
def test_ragged_data_times_continuous_adjoint(gpu, monkeypatch, fused):
    """ContinuousAdjoint: H_itp interpolates the glacier's own snapshots, the loss callbacks fire at its own data times
    (gradient.jl:287, :331-365); the batch == each glacier alone == the oracle."""
    sched_env(monkeypatch, ADJ_FUSED=fused)
    ph, shapes, own, om, gm, th_true, th0, gls, mbs = _two_glaciers(gpu)
    union = sorted(set(own[0]) | set(own[1]))
    refs, Lo, go, lam0, sto = [], [], [], [], []
    adj = O.ContinuousAdjointCfg(n_quadrature=16)
    for i in range(2):
        cfg = O.SimConfig(tstops=own[i], reltol=1e-8)
        ref, _, _ = O.forward(gls[i], O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th_true, T=-2.0), cfg)
        refs.append(ref)
        l, g, l0, s_ = O.loss_and_grad_continuous(gls[i], O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th0, T=-2.0), cfg, ref, own[i], adj)
        Lo.append(l); go.append(g); lam0.append(l0); sto.append(s_)
    b = _make_batch(gpu, [0, 1], shapes, gls, gm, th0, own, refs, mbs)
    for k in range(2):
        b.set_glacier_stops(k, own[k])
    Lb, gb = b.loss_grad_continuous(union, theta=th0, reltol=1e-8, n_quadrature=16)
    lam_b = [b.lambda0(k) for k in range(2)]
    srev = list(b.last_stats_rev)
    b.close()
    assert abs(Lb - sum(Lo)) <= 1e-6 * abs(sum(Lo))
    ratio, angle, relerr = stats_err_arrays(gb, go[0] + go[1])
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (ratio, angle, relerr)
    g_alone = np.zeros_like(gb)
    for i in range(2):
        assert rel_l2(lam_b[i], lam0[i]) < 1e-5
        assert abs(srev[i].naccept - sto[i].naccept) <= 2 and abs(srev[i].nreject - sto[i].nreject) <= 2, (i, srev[i], sto[i])
        b1 = _make_batch(gpu, [i], shapes, gls, gm, th0, own, refs, mbs)
        L1, g1 = b1.loss_grad_continuous(own[i], theta=th0, reltol=1e-8, n_quadrature=16)
        g_alone += g1
        assert rel_l2(b1.lambda0(0), lam_b[i]) < 1e-7
        b1.close()
    assert rel_l2(gb, g_alone) < 1e-7


def test_mass_balance_steps_that_are_not_result_stops(gpu):
    """run!(Prediction) with step_MB = 1/24 and solver.step = 1/12: the PeriodicCallback makes the integrator land on the
    half-month marks and apply the mass balance there; the result holds the monthly stops only."""
    nx, ny = 96, 80
    H0, B = O.synthetic_valley(nx, ny, 50.0)
    H0 = np.asfortranarray(H0); B = np.asfortranarray(B)
    step_mb, step = 1.0 / 24.0, 1.0 / 12.0
    p = gpu.Parameters(simulation=gpu.SimulationParameters(tspan=(2010.0, 2010.5), use_MB=True, step_MB=step_mb),
                       solver=gpu.SolverParameters(reltol=1e-8, step=step))
    S0 = B + H0
    ela = float(np.percentile(S0[H0 > 0], 55))
    mbm = gpu.LinearMB(grad=6e-3, ELA=ela, max_acc=1.2)
    A0 = 2.21e-18
    gl = [gpu.Glacier2D("valley", H0, B, 50.0, 50.0, A=A0)]
    pred = gpu.Prediction(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.ConstantA(A0)), mass_balance=mbm), gl, p)
    res = gpu.run_b(pred)[0]
    mbt = pred.mb_times()
    assert len(res.t) == 7 and len(mbt) == 12 and res.t[-1] == 2010.5
    assert sum(1 for t in mbt if t not in res.t) == 6  # six integrator stops that are not result stops
    mb = O.MassBalance(mb0=6e-3 * (S0 - ela) * step_mb, dmb_dS=6e-3 * step_mb, S_ref=S0, mb_max=1.2 * step_mb)
    cfg = O.SimConfig(tstops=res.t, reltol=1e-8, mb=mb, mb_times=mbt)
    snaps, st, _ = O.forward(O.Glacier(H0, B, 50.0, 50.0, O.Phys()), O.Law(kind=O.LAW_CONST_A, A=A0), cfg)
    assert len(snaps) == 7
    for j in range(1, 7):
        assert rel_l2(res.H[j], snaps[j]) < 1e-6, j
    assert abs(res.stats.naccept - st.naccept) <= 3
    # ... and the half-month applications are really there: without them half of the mass balance is missing
    cfg12 = O.SimConfig(tstops=res.t, reltol=1e-8, mb=mb, mb_times=res.t[1:])
    s12, _, _ = O.forward(O.Glacier(H0, B, 50.0, 50.0, O.Phys()), O.Law(kind=O.LAW_CONST_A, A=A0), cfg12)
    assert rel_l2(res.H[-1], s12[-1]) > 1e-4


@pytest.mark.parametrize("scheme", [1, 2])
def test_mass_balance_only_stops_in_a_ragged_batch(gpu, scheme):
    """Low level: two glaciers, one with a mass balance every 1/48 yr (not among anybody's stops), one without any -- the
    second one's integrator must not notice.  Fixed dt: == oracle to 1e-11 under both kernel schedules."""
    ph = O.Phys()
    shapes = [(64, 48), (48, 40)]
    ts = [T0, T0 + 1.0 / 24.0, T0 + 2.0 / 24.0]
    mbt = [T0 + (m + 1) / 48.0 for m in range(4)]  # 1/48, 2/48 (= ts[1]), 3/48, 4/48 (= ts[2])
    dt = 1.0 / 480.0
    b = gpu.GlacierBatch(shapes, [50.0, 50.0], A=[2.21e-18, 3e-18])
    gls, mbs = [], []
    for k, (nx, ny) in enumerate(shapes):
        H0, B = O.synthetic_valley(nx, ny, 50.0)
        gls.append(O.Glacier(H0, B, 50.0, 50.0, ph))
        b.set_fields(k, H0, B)
        mbs.append(_mb(H0, B, step=1.0 / 48.0) if k == 0 else None)
    b.set_mass_balance(0, mbs[0].mb0, mbs[0].dmb_dS, mbs[0].S_ref, mbs[0].mb_max)
    st = b.solve(ts, mb_times=mbt, fixed_dt=dt, scheme=scheme)
    for k, A in enumerate([2.21e-18, 3e-18]):
        cfg = O.SimConfig(tstops=ts, fixed_dt=dt, mb=mbs[k], mb_times=mbt if k == 0 else ())
        fo, so, _ = O.forward(gls[k], O.Law(kind=O.LAW_CONST_A, A=A), cfg)
        for j in range(3):
            assert rel_l2(b.snapshot(k, j), fo[j]) < 1e-11, (k, j)
        assert st[k].naccept == so.naccept
    # gradient entry points: the DiscreteAdjoint rejects such times like the reference (gradient.jl:131)
    b.set_reference(0, ts, [gls[0].H0] * 3, 3)
    b.set_reference(1, ts, [gls[1].H0] * 3, 3)
    with pytest.raises(Exception, match="MB callback"):
        b.loss_grad(ts, mb_times=mbt, fixed_dt=dt)
    b.close()


@pytest.mark.parametrize("adjoint", ["discrete", "continuous"])
def test_inversion_api_builds_the_stop_table_per_glacier(gpu, adjoint):
    """Inversion / SIA2D_grad_b on two glaciers whose thicknessData.t differ: every glacier gets ITS table (the shared `step`
    grid + its own data times, inversion_utils.jl:487-495) and loss / gradient equal the oracle's per-glacier sums."""
    step = 1.0 / 48.0
    p = gpu.Parameters(simulation=gpu.SimulationParameters(tspan=(2010.0, 2010.0 + 4 * step)),
                       solver=gpu.SolverParameters(reltol=1e-9, step=2 * step))
    nq = 12
    p.UDE.grad = gpu.DiscreteAdjoint() if adjoint == "discrete" else gpu.ContinuousAdjoint(n_quadrature=nq)
    grid = [2010.0 + 2 * j * step for j in range(3)]
    tH = [[grid[0], 2010.0 + step, grid[2]], [grid[0], grid[1], 2010.0 + 3.5 * step, grid[2]]]
    ph = O.Phys()
    gl = []
    for kk, (nx, ny) in enumerate([(48, 40), (64, 48)]):
        H0, B = O.synthetic_alpine(nx, ny, hmax=160.0, slope=0.1)
        g = gpu.Glacier2D(f"SYN-{kk}", H0, B, 50.0, 50.0, A=3e-17)
        g.thicknessData = gpu.ThicknessData(tH[kk], [H0 * (1.0 - 0.01 * j) for j in range(len(tH[kk]))])
        gl.append(g)
    reg = gpu.GlacierWideInv(p, gl, "A")
    inv = gpu.Inversion(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.LawA(p, scalar=True)), regressors={"A": reg}), gl, p)
    own = [inv.tstops_glacier(k) for k in range(2)]
    assert own[0] == sorted(set(grid) | set(tH[0])) and own[1] == sorted(set(grid) | set(tH[1])) and own[0] != own[1]
    assert inv.tstops() == sorted(set(own[0]) | set(own[1]))
    th = reg.theta.copy()
    dth = np.zeros_like(th)
    L = gpu.SIA2D_grad_b(dth, th, inv)
    lo, hi = ph.minA, ph.maxA
    Lo, go = 0.0, np.zeros(2)
    for kk, g in enumerate(gl):
        A = lo + (hi - lo) * (np.tanh(th[kk]) + 1) / 2
        cfg = O.SimConfig(tstops=own[kk], reltol=1e-9)
        og, law = O.Glacier(g.H0, g.B, 50.0, 50.0, ph), O.Law(kind=O.LAW_CONST_A, A=A)
        if adjoint == "discrete":
            l1, g1, _ = O.loss_and_grad(og, law, cfg, g.thicknessData.H, tH[kk])
        else:
            l1, g1, _, _ = O.loss_and_grad_continuous(og, law, cfg, g.thicknessData.H, tH[kk], O.ContinuousAdjointCfg(n_quadrature=nq))
        Lo += l1
        go[kk] = g1[0] * (hi - lo) / 2 * (1 - np.tanh(th[kk]) ** 2)
    assert abs(L - Lo) <= 1e-6 * abs(Lo)
    assert np.allclose(dth, go, rtol=2e-5)
    res = gpu.run_b(gpu.Prediction(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.ConstantA(3e-17))), gl, p))
    assert [r.t for r in res] == own and [len(r.H) for r in res] == [len(o) for o in own]


@pytest.mark.parametrize("fused", ["1", "0"])
def test_continuous_adjoint_with_mass_balance_steps_that_are_not_result_stops(gpu, monkeypatch, fused):
    """ContinuousAdjoint with step_MB = 1/48 yr and result stops every 1/24 yr: the reverse PeriodicCallback (gradient.jl:413-432)
    stops the reverse integrator at the in-between mass-balance times too and adds VJP_MB(lambda, H_itp(t) - MB_t), H_itp being
    the interpolant of the RESULT snapshots -- against the oracle's restatement."""
    sched_env(monkeypatch, ADJ_FUSED=fused)
    ph = O.Phys()
    nx, ny = 64, 48
    H0, B = O.synthetic_valley(nx, ny, 50.0)
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    ts = [T0 + j / 24.0 for j in range(4)]
    mbt = [T0 + (m + 1) / 48.0 for m in range(6)]
    mb = _mb(H0, B, step=1.0 / 48.0)
    om = O.default_nn(1, light=False, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    gm = gpu.MLPSpec(om.widths, om.acts, None, O.POST_AFFINE, ph.minA, ph.maxA)
    th_true, th0 = om.init_theta(np.random.default_rng(42)), om.init_theta(np.random.default_rng(1234))
    cfg = O.SimConfig(tstops=ts, reltol=1e-8, mb=mb, mb_times=mbt)
    ref, _, _ = O.forward(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th_true, T=-2.0), cfg)
    law0 = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th0, T=-2.0)
    adj = O.ContinuousAdjointCfg(n_quadrature=10)
    Lo, go, lam0, st_o = O.loss_and_grad_continuous(gl, law0, cfg, ref, ts, adj)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-2.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_A_SCALAR, gm, th0)
    b.set_reference(0, ts, ref, 3)
    b.set_mass_balance(0, mb.mb0, mb.dmb_dS, mb.S_ref, mb.mb_max)
    Lg, gg = b.loss_grad_continuous(ts, theta=th0, mb_times=mbt, reltol=1e-8, n_quadrature=10)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (ratio, angle, relerr)
    assert rel_l2(b.lambda0(0), lam0) < 1e-5
    sr = b.last_stats_rev[0]
    assert abs(sr.naccept - st_o.naccept) <= 2 and abs(sr.nreject - st_o.nreject) <= 2, (sr, st_o)
    # without the in-between applications the gradient is another one
    Lh, gh = b.loss_grad_continuous(ts, theta=th0, mb_times=ts[1:], reltol=1e-8, n_quadrature=10)
    assert rel_l2(gh, gg) > 1e-4
    b.close()
