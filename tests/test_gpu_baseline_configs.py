"""GPU: the five BASELINE.json configs as concrete synthetic inputs, each checked
against the oracle at the north-star bar (rel-L2 of H <= 1e-6) or, at full size, against the
oracle's C restatement on the same arithmetic sequence.  configs[3] (4-glacier inversion gradient)
lives in test_gpu_golden_api.py::test_config4_four_glaciers_gradient."""
import numpy as np
import pytest

from conftest import rel_l2, sched_env
from oracle import c_oracle as CO
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu
# the C oracle's parallel-for stages synchronise every stage: keep its team small (a 128-thread team on
# a quota-limited box spends its time in barriers)
CO.lib().oc_set_threads(min(8, CO.lib().oc_num_threads()))
A0 = 2.21e-18  # the reference's constant-A test value (test/grad_free_test.jl:44)


def argentiere_standin(nx=192, ny=160, dx=50.0):
    """configs[0] stand-in (real RGI60-11.03638 data is absent): valley bed
    B = 2200 - 0.12 x + 300 ((y-yc)/yhalf)^2, parabolic tongue <= 250 m inside an ellipse."""
    x = (np.arange(nx) * dx)[:, None]
    y = (np.arange(ny) * dx)[None, :]
    yc = ny * dx / 2
    B = 2200.0 - 0.12 * x + 300.0 * ((y - yc) / yc) ** 2
    ell = ((x - 0.5 * nx * dx) / (0.42 * nx * dx)) ** 2 + ((y - yc) / (0.22 * ny * dx)) ** 2
    H0 = np.maximum(0.0, 250.0 * (1.0 - ell))
    return np.asfortranarray(H0), np.asfortranarray(B + 0.0 * H0)


def test_config0_single_glacier_prediction_with_mass_balance(gpu):
    """configs[0]: SIA2Dmodel + TImodel1-like MB (linear elevation profile, DDF 6e-3, acc cap 1.2 m/yr),
    tspan (2010, 2015), monthly MB and snapshots: run!(Prediction) vs the oracle's forward."""
    H0, B = argentiere_standin()
    p = gpu.Parameters(simulation=gpu.SimulationParameters(tspan=(2010.0, 2015.0), use_MB=True, step_MB=1.0 / 12.0),
                       solver=gpu.SolverParameters(reltol=1e-8, step=1.0 / 12.0))
    S0 = B + H0
    ela = float(np.percentile(S0[H0 > 0], 55))
    mbm = gpu.LinearMB(grad=6e-3, ELA=ela, max_acc=1.2)
    gl = [gpu.Glacier2D("RGI60-11.03638-standin", H0, B, 50.0, 50.0, A=A0)]
    res = gpu.run_b(gpu.Prediction(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.ConstantA(A0)), mass_balance=mbm), gl, p))[0]
    assert len(res.t) == 61 and res.t[0] == 2010.0 and res.t[-1] == 2015.0
    step = 1.0 / 12.0
    mb = O.MassBalance(mb0=6e-3 * (S0 - ela) * step, dmb_dS=6e-3 * step, S_ref=S0, mb_max=1.2 * step)
    cfg = O.SimConfig(tstops=res.t, reltol=1e-8, mb=mb, mb_times=res.t[1:])
    snaps, st, _ = O.forward(O.Glacier(H0, B, 50.0, 50.0, O.Phys()), O.Law(kind=O.LAW_CONST_A, A=A0), cfg)
    for j in (1, 12, 30, 60):
        assert rel_l2(res.H[j], snaps[j]) < 1e-6, j
    assert abs(res.stats.naccept - st.naccept) <= max(3, st.naccept // 50)
    assert res.H[-1].min() >= 0.0 and res.H[-1].max() < 400.0


def _icecap(n, seed=None):
    H0, B = O.synthetic_icecap(n, n, 100.0)
    return np.asfortranarray(H0), np.asfortranarray(B)


def test_config1_512_icecap_forward(gpu):
    """configs[1]: 512^2, dx = 100 m, constant A, no MB.  (a) RHS == C oracle; (b) 12 fixed RK steps ==
    the C oracle's same arithmetic sequence; (c) one simulated year, adaptive: volume conserved,
    thickness stays >= 0, the device-side controller lands on the 12 monthly stops."""
    n = 512
    H0, B = _icecap(n)
    ph = O.Phys()
    b = gpu.GlacierBatch([(n, n)], [100.0], A=[A0])
    b.set_fields(0, H0, B)
    assert rel_l2(b.dhdt(0, H0), CO.rhs(H0, B, 100.0, 100.0, ph, A0)) < 1e-12
    st = CO.Stepper(H0, B, 100.0, 100.0, ph, A0)
    for _ in range(12):
        st.step(0.01)
    b.solve([0.0, 0.12], fixed_dt=0.01)
    assert rel_l2(b.snapshot(0, 1), st.u) < 1e-12
    ts = [2010.0 + k / 12.0 for k in range(13)]
    stats = b.solve(ts, reltol=1e-8)
    H1 = b.snapshot(0, 12)
    assert stats[0].t_final == ts[-1] and stats[0].naccept >= 12
    assert abs(H1.sum() - H0.sum()) <= 1e-11 * H0.sum() and H1.min() >= 0.0
    b.close()


def test_config2_512_icecap_with_nn_laws(gpu):
    """configs[2]: the same grid with a 2-hidden-layer x 16-unit MLP (softplus / sigmoid):
    (i) A = NN(T) on the gridded temperature T = -5 - 6.5e-3 (S - mean S), hoisted;
    (ii) Y = NN(T, Hbar) inlined per dual node, prescale (-25,0),(0,500).  RHS and one year forward."""
    n = 512
    H0, B = _icecap(n)
    ph = O.Phys()
    rng = np.random.default_rng(1234)
    S = B + H0
    Tg = -5.0 - 6.5e-3 * (O.avg(S) - S.mean())
    mA = O.MLP([1, 16, 16, 1], [O.ACT_SOFTPLUS, O.ACT_SOFTPLUS, O.ACT_SIGMOID], None, O.POST_AFFINE, ph.minA, ph.maxA)
    thA = mA.init_theta(rng)
    b = gpu.GlacierBatch([(n, n)], [100.0])
    b.set_fields(0, H0, B)
    b.set_T_field(0, Tg)
    b.set_law(gpu.LAW_NN_A_GRIDDED, gpu.MLPSpec(mA.widths, mA.acts, None, O.POST_AFFINE, ph.minA, ph.maxA), thA)
    lawA = O.Law(kind=O.LAW_NN_A_GRIDDED, mlp=mA, theta=thA, T=Tg)
    assert rel_l2(b.dhdt(0, H0), O.sia2d_rhs(H0, B, 100.0, 100.0, ph, lawA)) < 1e-11
    mY = O.MLP([2, 16, 16, 1], [O.ACT_SOFTPLUS, O.ACT_SOFTPLUS, O.ACT_SIGMOID], ((-25.0, 0.0), (0.0, 500.0)),
               O.POST_EXPMAX, 0.0, ph.maxA)
    thY = rng.uniform(-0.5, 0.5, mY.n_params)
    b.set_law(gpu.LAW_NN_Y, gpu.MLPSpec(mY.widths, mY.acts, mY.prescale, O.POST_EXPMAX, 0.0, ph.maxA), thY)
    lawY = O.Law(kind=O.LAW_NN_Y, mlp=mY, theta=thY, T=-5.0)
    b2 = gpu.GlacierBatch([(n, n)], [100.0], T=[-5.0])
    b2.set_fields(0, H0, B)
    b2.set_law(gpu.LAW_NN_Y, gpu.MLPSpec(mY.widths, mY.acts, mY.prescale, O.POST_EXPMAX, 0.0, ph.maxA), thY)
    assert rel_l2(b2.dhdt(0, H0), O.sia2d_rhs(H0, B, 100.0, 100.0, ph, lawY)) < 1e-10
    ts = [2010.0, 2010.5, 2011.0]
    st = b2.solve(ts, reltol=1e-8)
    H1 = b2.snapshot(0, 2)
    assert st[0].t_final == 2011.0 and abs(H1.sum() - H0.sum()) <= 1e-11 * H0.sum() and H1.min() >= 0.0
    b.close()
    b2.close()


def test_config2_512_forward_ude_and_adjoint_seams_match_the_oracle_at_full_size(gpu):
    """configs[2] at its FULL size against the oracle, not only through properties: 512^2, the 2 x 16 network,
    (i) A = NN(T) gridded and (ii) Y = NN(T, Hbar) inlined per dual node -- four fixed RDPK3Sp35 steps of the forward UDE
    (every kernel form the solve would pick at this size runs the same arithmetic) and the two VJP seams of the gradient
    (J_H^T lambda, J_theta^T lambda) on the stepped state, against the numpy oracle on the same inputs."""
    n = 512
    H0, B = _icecap(n)
    ph = O.Phys()
    rng = np.random.default_rng(1234)
    S = B + H0
    Tg = -5.0 - 6.5e-3 * (O.avg(S) - S.mean())
    mA = O.MLP([1, 16, 16, 1], [O.ACT_SOFTPLUS, O.ACT_SOFTPLUS, O.ACT_SIGMOID], None, O.POST_AFFINE, ph.minA, ph.maxA)
    thA = mA.init_theta(rng)
    mY = O.MLP([2, 16, 16, 1], [O.ACT_SOFTPLUS, O.ACT_SOFTPLUS, O.ACT_SIGMOID], ((-25.0, 0.0), (0.0, 500.0)),
               O.POST_EXPMAX, 0.0, ph.maxA)
    thY = rng.uniform(-0.5, 0.5, mY.n_params)
    lam = rng.standard_normal((n, n))
    dt, nsteps = 1e-4, 4  # (below the explicit stability limit of the largest A the random network can return, ~7e-4 yr)
    for name, kind, m, th, law, tol in (
            ("A(T) gridded", gpu.LAW_NN_A_GRIDDED, mA, thA, O.Law(kind=O.LAW_NN_A_GRIDDED, mlp=mA, theta=thA, T=Tg), 1e-11),
            ("Y(T, Hbar) inlined", gpu.LAW_NN_Y, mY, thY, O.Law(kind=O.LAW_NN_Y, mlp=mY, theta=thY, T=-5.0, interpolation="none"), 1e-10)):
        b = gpu.GlacierBatch([(n, n)], [100.0], T=[-5.0])
        b.set_fields(0, H0, B)
        if kind == gpu.LAW_NN_A_GRIDDED:
            b.set_T_field(0, Tg)
        b.set_law(kind, gpu.MLPSpec(m.widths, m.acts, m.prescale, m.post_kind, m.post_lo, m.post_hi), th)
        if kind == gpu.LAW_NN_Y:
            b.set_grad_interpolation(gpu._lib.GRAD_INTERP_NONE, 75)
        f = lambda H: O.sia2d_rhs(H, B, 100.0, 100.0, ph, law)
        u = H0
        for _ in range(nsteps):
            u, _ = O.rdpk3sp35_step(f, u, dt)
        b.solve([0.0, dt * nsteps], fixed_dt=dt)
        H1 = b.snapshot(0, 1)
        assert rel_l2(H1, u) < 1e-12, name
        assert rel_l2(b.vjp_H(0, lam, H1), O.vjp_H(lam, u, B, 100.0, 100.0, ph, law)) < (1e-10 if kind == gpu.LAW_NN_A_GRIDDED else 1e-8), name
        assert rel_l2(b.vjp_theta(0, lam, H1), O.vjp_theta(lam, u, B, 100.0, 100.0, ph, law)) < tol * 10, name
        b.close()


def test_config4_batch_of_1024_glaciers(gpu, monkeypatch):
    """configs[4], per-GPU share: 8 caps at 1024^2 with per-glacier random (R, bed phase, A); one fused
    RDPK3Sp35 step sequence of every glacier == the C oracle stepping that glacier alone, and the
    batch result does not depend on the batch composition: bit for bit when the same step kernel runs
    (the library picks the fused kernel by batch size -- here the 8-row strip kernel for the batch, the
    7-row one for a single glacier --, which changes results at rounding level only)."""
    import bench

    sched_env(monkeypatch, FUSED_TILES=None)  # this test is about the library's own choice
    n, G = 1024, 8
    gl = [bench.make_glacier(n, k) for k in range(G)]
    b = gpu.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl])
    for k, (H0, B, A) in enumerate(gl):
        b.set_fields(k, H0, B)
    b.solve([0.0, 0.02], fixed_dt=0.005)
    ph = O.Phys()
    for k in (0, 3, 7):
        H0, B, A = gl[k]
        st = CO.Stepper(H0, B, 100.0, 100.0, ph, A)
        for _ in range(4):
            st.step(0.005)
        assert rel_l2(b.snapshot(k, 1), st.u) < 1e-12, k
    one = gpu.GlacierBatch([(n, n)], [100.0], A=[gl[5][2]])
    one.set_fields(0, gl[5][0], gl[5][1])
    one.solve([0.0, 0.02], fixed_dt=0.005)
    assert rel_l2(one.snapshot(0, 1), b.snapshot(5, 1)) < 1e-13
    one.close()
    sched_env(monkeypatch, FUSED_TILES="u")  # the kernel the batch of 8 ran on ...
    sched_env(monkeypatch, STEP_SC="0")      # ... in the same instantiation (three-launch loop)
    one = gpu.GlacierBatch([(n, n)], [100.0], A=[gl[5][2]])
    one.set_fields(0, gl[5][0], gl[5][1])
    one.solve([0.0, 0.02], fixed_dt=0.005)
    assert np.array_equal(one.snapshot(0, 1), b.snapshot(5, 1))
    one.close()
    b.close()
