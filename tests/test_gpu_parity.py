"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on
identical seeded inputs.  fp64 tolerances are written in each test; the north-star bar is
rel-L2 <= 1e-6 on the final H and the reference's FD thresholds on the gradients."""
import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays, sched_env
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu

RHS_TOL = 1e-12  # one RHS / VJP evaluation, fp64, rel-L2
A0 = 2.21e-18


def _setup(odinn, nx, ny, dx=100.0, scale=0.4, phys=None, A=A0, valley=False):
    if valley:
        H0, B = O.synthetic_valley(nx, ny, dx)
    else:
        H0, B = O.synthetic_icecap(nx, ny, dx)
        H0 = H0 * scale
    ph = phys or O.Phys()
    b = odinn.GlacierBatch([(nx, ny)], [dx], phys=[odinn.PhysicalParameters(**ph.__dict__)], A=[A])
    b.set_fields(0, H0, B)
    return b, H0, B, ph


def _mlp_pair(odinn, widths, acts, prescale, post_kind, lo, hi, seed=1234):
    om = O.MLP(widths, acts, prescale, post_kind, lo, hi)
    th = om.init_theta(np.random.default_rng(seed))
    # non-zero biases so that the bias gradient paths are exercised
    th = th + 0.05 * np.random.default_rng(seed + 1).standard_normal(th.size)
    gm = odinn.MLPSpec(widths, acts, prescale, post_kind, lo, hi)
    return om, gm, th


@pytest.mark.parametrize("shape", [(96, 80), (64, 16), (65, 17), (130, 50), (37, 40), (3, 3), (200, 131)])
def test_dhdt_constA(gpu, shape):
    nx, ny = shape
    b, H0, B, ph = _setup(gpu, nx, ny)
    law = O.Law(kind=O.LAW_CONST_A, A=A0)
    ref = O.sia2d_rhs(H0, B, 100.0, 100.0, ph, law)
    got = b.dhdt(0, H0)
    if np.linalg.norm(ref) == 0:
        assert np.all(got == 0)
    else:
        assert rel_l2(got, ref) < RHS_TOL
    assert np.all(got[0, :] == 0) and np.all(got[-1, :] == 0) and np.all(got[:, 0] == 0) and np.all(got[:, -1] == 0)
    b.close()


def test_dhdt_negative_and_random_H(gpu):
    """max(H,0) clamp (adjoint.jl:52) and active border clamps on rough random input."""
    nx, ny = 70, 45
    rng = np.random.default_rng(1234)
    b, H0, B, ph = _setup(gpu, nx, ny)
    H = np.asfortranarray(np.abs(rng.standard_normal((nx, ny))) * 50.0 - 10.0)
    B2 = np.asfortranarray(B + 30.0 * rng.standard_normal((nx, ny)))
    b.set_fields(0, H, B2)
    ref = O.sia2d_rhs(H, B2, 100.0, 100.0, ph, O.Law(kind=O.LAW_CONST_A, A=A0))
    assert rel_l2(b.dhdt(0, H), ref) < RHS_TOL
    b.close()


def test_dhdt_generic_exponents_and_sliding(gpu):
    """pow path: n != 3 and C > 0 (reference tests C = 7e-8, runtests.jl:92-94)."""
    for n, C, eta0 in [(3.0, 7e-8, 1.0), (2.6, 0.0, 1.0), (3.3, 7e-8, 1.0), (3.0, 0.0, 0.7)]:
        ph = O.Phys(n=n, C=C, p=3.0, q=1.0, eta0=eta0)
        b, H0, B, _ = _setup(gpu, 80, 60, phys=ph)
        law = O.Law(kind=O.LAW_CONST_A, A=A0)
        ref = O.sia2d_rhs(H0, B, 100.0, 100.0, ph, law)
        assert rel_l2(b.dhdt(0, H0), ref) < 1e-11
        lam = np.random.default_rng(7).standard_normal(H0.shape)
        assert rel_l2(b.vjp_H(0, lam, H0), O.vjp_H(lam, H0, B, 100.0, 100.0, ph, law)) < 1e-10
        assert rel_l2(b.vjp_theta(0, lam, H0), O.vjp_theta(lam, H0, B, 100.0, 100.0, ph, law)) < 1e-10
        b.close()


@pytest.mark.parametrize("shape", [(96, 80), (65, 17), (37, 40), (3, 3), (130, 50)])
def test_vjp_H_and_theta_constA(gpu, shape):
    nx, ny = shape
    b, H0, B, ph = _setup(gpu, nx, ny)
    rng = np.random.default_rng(1234)
    lam = rng.standard_normal((nx, ny))
    law = O.Law(kind=O.LAW_CONST_A, A=A0)
    ref = O.vjp_H(lam, H0, B, 100.0, 100.0, ph, law)
    got = b.vjp_H(0, lam, H0)
    if np.linalg.norm(ref) > 0:
        assert rel_l2(got, ref) < 1e-11
        rt = O.vjp_theta(lam, H0, B, 100.0, 100.0, ph, law)
        gt = b.vjp_theta(0, lam, H0)
        assert abs(gt[0] - rt[0]) <= 1e-11 * abs(rt[0])
    else:
        assert np.all(got == 0)
    b.close()


def test_vjp_flat_bed_ties(gpu):
    """Halfar-like dome on a flat bed: structural ties e == eta0*H/dx at the margin must take
    the same (strict-inequality) branch as the reference (inversion_utils.jl:24-28)."""
    nx = ny = 60
    dx = 2000.0 / nx / 0.4
    x = (np.arange(nx) - nx / 2 + 0.5) * dx
    X, Y = np.meshgrid(x, x, indexing="ij")
    H = np.asfortranarray(O.halfar(X, Y, O.halfar_t0(1.1e-17, 400.0, 2000.0), 1.1e-17, 400.0, 2000.0))
    B = np.zeros_like(H)
    ph = O.Phys()
    b = gpu.GlacierBatch([(nx, ny)], [dx], A=[1.1e-17])
    b.set_fields(0, H, B)
    law = O.Law(kind=O.LAW_CONST_A, A=1.1e-17)
    lam = np.random.default_rng(3).standard_normal((nx, ny))
    assert rel_l2(b.dhdt(0, H), O.sia2d_rhs(H, B, dx, dx, ph, law)) < RHS_TOL
    assert rel_l2(b.vjp_H(0, lam, H), O.vjp_H(lam, H, B, dx, dx, ph, law)) < 1e-11
    b.close()


def test_vjp_meets_reference_fd_thresholds(gpu):
    """The reference's own acceptance test for the discrete VJP (test/SIA2D_adjoint.jl,
    thresholds [5e-7, 1e-6, 5e-4] runtests.jl:89-91), with the GPU RHS as the function
    being differenced and the GPU VJP as the candidate."""
    nx, ny = 40, 37
    b, H0, B, ph = _setup(gpu, nx, ny, scale=0.3)
    rng = np.random.default_rng(1234)
    lam = rng.standard_normal((nx, ny))
    g = b.vjp_H(0, lam, H0)
    best = [np.inf] * 3
    f0 = np.sum(b.dhdt(0, H0) * lam)
    for eps in (1e-3, 1e-5, 1e-7):
        gn = np.zeros_like(H0)
        for i in range(nx):
            for j in range(ny):
                Hp = H0.copy()
                Hp[i, j] += eps
                gn[i, j] = (np.sum(b.dhdt(0, Hp) * lam) - f0) / eps
        st = stats_err_arrays(g, gn)
        best = [min(a, abs(s)) for a, s in zip(best, st)]
    assert best[0] < 5e-7 and best[1] < 1e-6 and best[2] < 5e-4, best
    b.close()


def test_nn_A_scalar(gpu):
    """LawA(nn; scalar=true): A hoisted (callback_freq = 0), dtheta = dA/dtheta * sum(spat*Da)."""
    ph = O.Phys()
    om, gm, th = _mlp_pair(gpu, [1, 3, 10, 3, 1], [1, 1, 1, 2], None, O.POST_AFFINE, ph.minA, ph.maxA)
    b, H0, B, _ = _setup(gpu, 96, 80)
    b.set_law(gpu.LAW_NN_A_SCALAR, gm, th)
    law = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th, T=-5.0)
    assert abs(b.eval_law(0) - O.law_value(law, ph, None, None)) <= 1e-13 * ph.maxA
    assert rel_l2(b.dhdt(0, H0), O.sia2d_rhs(H0, B, 100.0, 100.0, ph, law)) < RHS_TOL
    lam = np.random.default_rng(5).standard_normal(H0.shape)
    assert rel_l2(b.vjp_H(0, lam, H0), O.vjp_H(lam, H0, B, 100.0, 100.0, ph, law)) < 1e-11
    assert rel_l2(b.vjp_theta(0, lam, H0), O.vjp_theta(lam, H0, B, 100.0, 100.0, ph, law)) < 1e-10
    b.close()


@pytest.mark.parametrize("wave", ["1", "0"])
@pytest.mark.parametrize("arch", ["default", "w16", "runtime", "wide"])
def test_nn_A_gridded(gpu, monkeypatch, arch, wave):
    """LawA(nn; scalar = false): A = NN(T) hoisted onto the dual grid; dtheta = sum_nodes Gacc dA/dtheta(T) by the wave-reduced
    backprop kernel (k_law_field_grad_wave: compile-time 1-3-10-3-1 and 1-16-16-1 nets, run-time architectures) and by the
    per-thread-accumulator kernel it replaces (ODINN_LAWGRAD_WAVE=0)."""
    sched_env(monkeypatch, LAWGRAD_WAVE=wave)
    ph = O.Phys()
    widths, acts = {
        "default": ([1, 3, 10, 3, 1], [1, 1, 1, 2]),
        "w16": ([1, 16, 16, 1], [1, 1, 2]),
        "runtime": ([1, 5, 7, 1], [3, 4, 2]),
        "wide": ([1, 20, 30, 1], [1, 1, 2]),
    }[arch]
    om, gm, th = _mlp_pair(gpu, widths, acts, None, O.POST_AFFINE, ph.minA, ph.maxA)
    nx, ny = 96, 80
    b, H0, B, _ = _setup(gpu, nx, ny)
    S = B + H0
    T = np.asfortranarray(-5.0 - 6.5e-3 * (O.avg(S) - S.mean()))
    b.set_law(gpu.LAW_NN_A_GRIDDED, gm, th)
    b.set_T_field(0, T)
    law = O.Law(kind=O.LAW_NN_A_GRIDDED, mlp=om, theta=th, T=T)
    assert rel_l2(b.eval_law(0, H0), O.law_value(law, ph, None, None)) < 1e-13
    assert rel_l2(b.dhdt(0, H0), O.sia2d_rhs(H0, B, 100.0, 100.0, ph, law)) < RHS_TOL
    lam = np.random.default_rng(5).standard_normal(H0.shape)
    assert rel_l2(b.vjp_H(0, lam, H0), O.vjp_H(lam, H0, B, 100.0, 100.0, ph, law)) < 1e-11
    assert rel_l2(b.vjp_theta(0, lam, H0), O.vjp_theta(lam, H0, B, 100.0, 100.0, ph, law)) < 1e-10
    b.close()


@pytest.mark.parametrize("arch", ["default", "w16", "light", "gelu5", "wide"])
def test_nn_Y_inlined(gpu, arch):
    """LawY: per-dual-node MLP(T, Hbar) inlined in the stencil (Laws.jl:258-265), :D_hybrid."""
    ph = O.Phys()
    widths, acts = {
        "default": ([2, 3, 10, 3, 1], [1, 1, 1, 2]),
        "w16": ([2, 16, 16, 1], [1, 1, 2]),
        "light": ([2, 3, 1], [1, 2]),
        "gelu5": ([2, 5, 10, 5, 1], [3, 3, 3, 1]),  # run-time architecture, <= 16 units (test/test_grad_loss.jl:182-190)
        "wide": ([2, 5, 8, 20, 30, 10, 1], [3, 3, 1, 1, 1, 2]),  # run-time, <= 32 units (scripts/MWEs/inversion_diffusivity)
    }[arch]
    om, gm, th = _mlp_pair(gpu, widths, acts, [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
    b, H0, B, _ = _setup(gpu, 80, 48)
    b.set_law(gpu.LAW_NN_Y, gm, th)
    law = O.Law(kind=O.LAW_NN_Y, mlp=om, theta=th, T=-5.0)
    Hc, S, gSx, gSy, gS, Hbar, *_ = O._forward_intermediates(H0, B, 100.0, 100.0, ph)
    assert rel_l2(b.eval_law(0, H0), O.law_value(law, ph, Hbar, gS)) < 1e-12
    assert rel_l2(b.dhdt(0, H0), O.sia2d_rhs(H0, B, 100.0, 100.0, ph, law)) < 1e-11
    lam = np.random.default_rng(5).standard_normal(H0.shape)
    # alpha contains a forward finite difference with dH = 1e-4 (target_D_hybrid.jl:58-71):
    # (a-b)/1e-4 amplifies ulp differences of exp/log by 1e4
    assert rel_l2(b.vjp_H(0, lam, H0), O.vjp_H(lam, H0, B, 100.0, 100.0, ph, law)) < 1e-8
    assert rel_l2(b.vjp_theta(0, lam, H0), O.vjp_theta(lam, H0, B, 100.0, 100.0, ph, law)) < 1e-10
    b.close()


@pytest.mark.parametrize("arch", ["default", "gelu5", "wide"])
def test_nn_U_inlined(gpu, arch):
    """LawU: D = Hbar * NN(Hbar, gradS) (Laws.jl:114-123), target :D.  gelu5 / wide: run-time architectures (the rolled
    evaluator over padded weight rows, 16 and 32 wide)."""
    ph = O.Phys()
    widths, acts = {"default": ([2, 3, 10, 3, 1], [1, 1, 1, 2]), "gelu5": ([2, 5, 10, 5, 1], [3, 3, 3, 1]),
                    "wide": ([2, 5, 8, 20, 30, 10, 1], [3, 3, 1, 1, 1, 2])}[arch]
    om, gm, th = _mlp_pair(gpu, widths, acts, [(0.0, 300.0), (0.0, 0.5)], O.POST_EXPMAX, 0.0, 50.0)
    b, H0, B, _ = _setup(gpu, 80, 48)
    b.set_law(gpu.LAW_NN_U, gm, th)
    law = O.Law(kind=O.LAW_NN_U, mlp=om, theta=th)
    assert rel_l2(b.dhdt(0, H0), O.sia2d_rhs(H0, B, 100.0, 100.0, ph, law)) < 1e-11
    lam = np.random.default_rng(5).standard_normal(H0.shape)
    # alpha, beta by central FD with 1e-4 / 1e-6 steps (target_D_pure.jl:105-137)
    assert rel_l2(b.vjp_H(0, lam, H0), O.vjp_H(lam, H0, B, 100.0, 100.0, ph, law)) < 1e-6
    assert rel_l2(b.vjp_theta(0, lam, H0), O.vjp_theta(lam, H0, B, 100.0, 100.0, ph, law)) < 1e-10
    b.close()


def _mb(H0, B, step=1.0 / 12.0):
    S0 = B + np.maximum(H0, 0.0)
    ela = np.percentile(S0[H0 > 0], 60)
    grad = 6e-3
    return O.MassBalance(mb0=grad * (S0 - ela) * step, dmb_dS=grad * step, S_ref=S0, mb_max=1.2 * step)


def test_mass_balance_seams(gpu):
    b, H0, B, ph = _setup(gpu, 96, 80)
    mb = _mb(H0, B)
    b.set_mass_balance(0, mb.mb0, mb.dmb_dS, mb.S_ref, mb.mb_max)
    rng = np.random.default_rng(11)
    H = np.asfortranarray(np.maximum(H0 + 5.0 * rng.standard_normal(H0.shape), 0.0))
    H[H0 > 0] = np.minimum(H[H0 > 0], 0.02 + H[H0 > 0] * (rng.random(np.sum(H0 > 0)) > 0.1))  # some nearly-empty cells
    Hn, MB = b.mb_apply(0, H)
    Hr, MBr = O.mb_apply(mb, H, B)
    # the device contracts mb0 + dmb_dS*(S - S_ref) into an FMA: agreement to 1 ulp, same mask/clip branches
    assert np.allclose(Hn, Hr, rtol=1e-14, atol=1e-15) and np.allclose(MB, MBr, rtol=1e-13, atol=1e-16)
    assert np.array_equal(MB == 0, MBr == 0) and np.array_equal(Hn == 0, Hr == 0)
    lam = rng.standard_normal(H.shape)
    assert np.allclose(b.mb_vjp_H(0, lam, H), O.vjp_mb(mb, lam, H, B), rtol=1e-14, atol=0)
    b.close()


def test_solve_fixed_dt_bit_level(gpu):
    """Non-adaptive RDPK3Sp35: same arithmetic sequence on both sides -> agreement to rounding."""
    b, H0, B, ph = _setup(gpu, 96, 80)
    law = O.Law(kind=O.LAW_CONST_A, A=A0)
    ts = [0.0, 0.1, 0.25]
    f = lambda H: O.sia2d_rhs(H, B, 100.0, 100.0, ph, law)
    snaps, st, _ = O.solve(f, H0, ts, fixed_dt=0.01)
    stats = b.solve(ts, fixed_dt=0.01)
    assert stats[0].naccept == st.naccept
    for j in range(3):
        assert rel_l2(b.snapshot(0, j), snaps[j]) < 1e-12
    b.close()


@pytest.mark.parametrize("reltol", [1e-8, 1e-6])
def test_solve_adaptive_matches_oracle(gpu, reltol):
    """Adaptive RDPK3Sp35 + PID, all control on the device, vs the same algorithm on the CPU.
    North-star bar: rel-L2(final H) <= 1e-6."""
    b, H0, B, ph = _setup(gpu, 96, 80)
    law = O.Law(kind=O.LAW_CONST_A, A=A0)
    ts = [2010.0 + k / 12.0 for k in range(7)]
    f = lambda H: O.sia2d_rhs(H, B, 100.0, 100.0, ph, law)
    snaps, st, _ = O.solve(f, H0, ts, reltol=reltol)
    stats = b.solve(ts, reltol=reltol)
    assert abs(stats[0].naccept - st.naccept) <= max(2, st.naccept // 50)
    for j in range(len(ts)):
        assert rel_l2(b.snapshot(0, j), snaps[j]) < 1e-6
    # volume conservation: no MB, ice does not touch the boundary ring
    assert abs(b.snapshot(0, len(ts) - 1).sum() - H0.sum()) <= 1e-12 * H0.sum()
    b.close()


def test_solve_with_mass_balance(gpu):
    b, H0, B, ph = _setup(gpu, 96, 80, valley=True, dx=50.0)
    mb = _mb(H0, B)
    b.set_mass_balance(0, mb.mb0, mb.dmb_dS, mb.S_ref, mb.mb_max)
    law = O.Law(kind=O.LAW_CONST_A, A=A0)
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    ts = [2010.0 + k / 12.0 for k in range(5)]
    cfg = O.SimConfig(tstops=ts, reltol=1e-8, mb=mb, mb_times=ts[1:])
    snaps, st, inc = O.forward(gl, law, cfg)
    b.solve(ts, mb_times=ts[1:], reltol=1e-8)
    for j in range(len(ts)):
        assert rel_l2(b.snapshot(0, j), snaps[j]) < 1e-6
    b.close()


def _inversion_case(odinn, nx, ny, use_mb=False, k=7, step=1.0 / 12.0):
    ph = O.Phys()
    H0, B = O.synthetic_valley(nx, ny, 50.0)
    ts = [2010.0 + j * step for j in range(k)]
    om = O.default_nn(1, light=False, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    gm = odinn.MLPSpec(om.widths, om.acts, None, O.POST_AFFINE, ph.minA, ph.maxA)
    th_true = om.init_theta(np.random.default_rng(42))
    th0 = om.init_theta(np.random.default_rng(1234))
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    mb = _mb(H0, B) if use_mb else None
    cfg = O.SimConfig(tstops=ts, reltol=1e-8, mb=mb, mb_times=ts[1:] if use_mb else ())
    ref_snaps, _, _ = O.forward(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th_true, T=-2.0), cfg)
    return ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref_snaps


@pytest.mark.parametrize("use_mb", [False, True])
def test_loss_grad_matches_oracle(gpu, use_mb):
    """odinn_loss_grad == SIA2D_grad_batch! (DiscreteAdjoint + DiscreteVJP) vs the oracle's
    restatement of gradient.jl:129-275 on the same inputs."""
    nx, ny = 64, 48
    ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref = _inversion_case(gpu, nx, ny, use_mb)
    law0 = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th0, T=-2.0)
    Lo, go, lam0 = O.loss_and_grad(gl, law0, cfg, ref, ts)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-2.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_A_SCALAR, gm, th0)
    b.set_reference(0, ts, ref, 3)
    if use_mb:
        b.set_mass_balance(0, mb.mb0, mb.dmb_dS, mb.S_ref, mb.mb_max)
    Lg, gg = b.loss_grad(ts, theta=th0, mb_times=ts[1:] if use_mb else (), reltol=1e-8)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (ratio, angle, relerr)
    assert rel_l2(b.lambda0(0), lam0) < 1e-5
    # forward-only loss agrees with the loss recomputed in the reverse loop (gradient.jl:259, rtol 1e-8)
    assert abs(b.loss()[0] - Lg) <= 1e-8 * abs(Lg)
    b.close()


def test_loss_grad_vs_finite_differences(gpu):
    """The reference's end-to-end acceptance test (test_grad_finite_diff, thresholds
    [5e-3, 1e-8, 5e-3] runtests.jl:116-117): dL/dtheta from the discrete adjoint vs central
    finite differences of the GPU forward loss."""
    nx, ny = 64, 48
    # snapshot spacing 1/240 yr: the reference's reverse scheme is explicit Euler at snapshot
    # resolution (gradient.jl:242) and is only first-order accurate -- at 1/12 yr it is unstable
    # on this fast synthetic valley (the reference warns about exactly that, gradient.jl:19-24)
    ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref = _inversion_case(gpu, nx, ny, False, k=13, step=1.0 / 240.0)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-2.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_A_SCALAR, gm, th0)
    b.set_reference(0, ts, ref, 3)
    L0, g = b.loss_grad(ts, theta=th0, reltol=1e-10)

    def loss_at(th):
        b.set_theta(th)
        b.solve(ts, reltol=1e-10)
        return b.loss()[0]

    # A = minA + (maxA-minA) sigmoid(...) makes L depend on theta through ONE scalar, so the
    # gradient direction is dA/dtheta: compare along it and along random directions
    gn = np.zeros_like(g)
    idx = np.arange(0, g.size, 5)
    for q in idx:
        e = np.zeros_like(g)
        e[q] = 1e-4
        gn[q] = (loss_at(th0 + e) - loss_at(th0 - e)) / (2 * e[q])
    ratio, angle, relerr = stats_err_arrays(g[idx], gn[idx])
    assert abs(ratio) < 5e-3 and abs(angle) < 1e-8 and relerr < 5e-3, (ratio, angle, relerr)
    b.close()


def test_batch_of_ragged_glaciers(gpu):
    """Several glaciers of different sizes in ONE batch (one launch per stage for all of them):
    every glacier must match its single-glacier oracle run; loss/grad are sums (Model.jl:208-224)."""
    shapes = [(96, 80), (64, 48), (37, 40), (130, 50)]
    ph = O.Phys()
    ts = [2010.0 + j / 12.0 for j in range(4)]
    Hs, Bs, As = [], [], [2.21e-18, 1e-17, 4e-18, 3e-17]
    for (nx, ny) in shapes:
        H0, B = O.synthetic_valley(nx, ny, 50.0)
        Hs.append(H0)
        Bs.append(B)
    b = gpu.GlacierBatch(shapes, [50.0] * 4, A=As)
    for g in range(4):
        b.set_fields(g, Hs[g], Bs[g])
    stats = b.solve(ts, reltol=1e-8)
    for g in range(4):
        law = O.Law(kind=O.LAW_CONST_A, A=As[g])
        f = lambda H, g=g, law=law: O.sia2d_rhs(H, Bs[g], 50.0, 50.0, ph, law)
        snaps, st, _ = O.solve(f, Hs[g], ts, reltol=1e-8)
        assert rel_l2(b.snapshot(g, 3), snaps[3]) < 1e-6
        assert rel_l2(b.dhdt(g, Hs[g]), f(Hs[g])) < RHS_TOL
    b.close()


def test_determinism(gpu):
    """Gather-form adjoint + fixed-order reductions: two runs are bitwise identical."""
    nx, ny = 64, 48
    ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref = _inversion_case(gpu, nx, ny, False, k=4)
    out = []
    for _ in range(2):
        b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-2.0])
        b.set_fields(0, H0, B)
        b.set_law(gpu.LAW_NN_A_SCALAR, gm, th0)
        b.set_reference(0, ts, ref, 3)
        out.append(b.loss_grad(ts, theta=th0, reltol=1e-8))
        b.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])


def test_halfar_known_answer(gpu):
    """Halfar similarity solution (reference set-up
    scripts/MWEs/inversion_diffusivity/inversion_setup.jl:44-68): dome height and volume."""
    nx = ny = 60
    R0, H00, A = 2000.0, 400.0, 1.1e-17
    dx = R0 / nx / 0.4
    x = (np.arange(nx) - nx / 2 + 0.5) * dx
    X, Y = np.meshgrid(x, x, indexing="ij")
    t0 = O.halfar_t0(A, H00, R0)
    H0 = np.asfortranarray(O.halfar(X, Y, t0, A, H00, R0))
    b = gpu.GlacierBatch([(nx, ny)], [dx], A=[A])
    b.set_fields(0, H0, np.zeros_like(H0))
    b.solve([t0, t0 + 10.0], reltol=1e-8)
    H1 = b.snapshot(0, 1)
    exact = O.halfar(X, Y, t0 + 10.0, A, H00, R0)
    assert abs(H1.max() - exact.max()) < 0.5  # metres; margin-resolution limited
    assert rel_l2(H1, exact) < 3e-2
    assert abs(H1.sum() - H0.sum()) < 1e-12 * H0.sum()
    b.close()


def test_full_size_properties(gpu):
    """BASELINE sizes (512^2, 1024^2): size-independent properties -- adjoint identity
    <lam, J v> == <J^T lam, v> through a directional FD, volume conservation of one RK step,
    zero boundary ring, linearity of the VJP in lambda."""
    for n in (512, 1024):
        H0, B = O.synthetic_icecap(n, n, 100.0)
        b = gpu.GlacierBatch([(n, n)], [100.0], A=[A0])
        b.set_fields(0, H0, B)
        rng = np.random.default_rng(n)
        lam = rng.standard_normal((n, n))
        v = rng.standard_normal((n, n)) * (H0 > 0)
        eps = 1e-4
        Jv = (b.dhdt(0, H0 + eps * v) - b.dhdt(0, H0 - eps * v)) / (2 * eps)
        lhs = np.sum(lam * Jv)
        rhs = np.sum(b.vjp_H(0, lam, H0) * v)
        assert abs(lhs - rhs) <= 1e-6 * abs(lhs)
        g1 = b.vjp_H(0, lam, H0)
        g2 = b.vjp_H(0, 2.0 * lam, H0)
        assert rel_l2(g2, 2.0 * g1) < 1e-14
        dH = b.dhdt(0, H0)
        assert abs(dH.sum()) <= 1e-10 * np.abs(dH).sum()  # flux form: interior divergence sums to ~0
        assert np.all(dH[0, :] == 0) and np.all(dH[:, -1] == 0)
        b.close()


@pytest.mark.parametrize("tiles", ["small", "large", "t", "u"])
def test_ice_free_tile_shortcut_is_bitwise_exact(gpu, monkeypatch, tiles):
    """The fused step kernels skip workgroups whose whole halo region has u == 0; the result must be
    bit-identical to running all five stages everywhere (opts.dense = 1).  tiles: the 54x8 latency tile,
    the 54x40 row-interleaved kernel, the 54x46 strip kernel ("t")."""
    sched_env(monkeypatch, FUSED_TILES=tiles)
    n = 320
    H0, B = O.synthetic_icecap(n, n, 100.0)
    H0 = np.where(H0 > 500.0, H0 - 500.0, 0.0)  # small cap: most tiles ice-free
    ts = [0.0, 0.5, 1.0]
    out = []
    for dense in (0, 1):
        b = gpu.GlacierBatch([(n, n)], [100.0], A=[4e-17])
        b.set_fields(0, np.asfortranarray(H0), B)
        st = b.solve(ts, reltol=1e-8, dense=dense)
        out.append((b.snapshot(0, 1), b.snapshot(0, 2), st[0].naccept, st[0].nreject))
        b.close()
    assert out[0][2:] == out[1][2:]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert (out[0][1] > 0).sum() > (H0 > 0).sum()  # the cap spread into previously ice-free cells


def test_fused_tile_sizes_are_equivalent(gpu, monkeypatch):
    """The fused step exists as a latency tile (54x8, picked when the batch cannot fill the GPU), a
    row-interleaved throughput kernel (54x40) and the strip kernel (54x46, "t": y-neighbours in
    registers, the default for the integer-power law on large batches): the same expression sequence
    per cell, compiled three times -- the solutions agree to rounding (the compiler is free to contract
    a*b + c*d either way per instantiation; the error norm is summed over other tile partials)."""
    H0, B = O.synthetic_valley(130, 97, 50.0)
    ts = [2010.0, 2010.25, 2010.5]
    out = {}
    for tiles in ("small", "large", "t", "u"):
        sched_env(monkeypatch, FUSED_TILES=tiles)
        b = gpu.GlacierBatch([(130, 97)], [50.0], A=[4e-17])
        b.set_fields(0, H0, B)
        st = b.solve(ts, reltol=1e-8)
        ad = (b.snapshot(0, 2), st[0].naccept, st[0].nreject)
        b.solve(ts, fixed_dt=0.002)
        out[tiles] = ad + (b.snapshot(0, 2),)
        b.close()
    # same dt sequence => equal to rounding.  Adaptive runs: the embedded error (u' - u) - utilde cancels ~6 digits,
    # so an ulp of difference in u' (the strip kernel is separate source: flux form, contracted differently) moves
    # the error norm in its 10th digit and the PID step sizes with it: agreement is bounded by the integration
    # error (~10 reltol), not by rounding.
    for other, tol in (("large", 1e-12), ("t", 1e-7), ("u", 1e-7)):
        assert out["small"][1:3] == out[other][1:3], other
        assert rel_l2(out["small"][0], out[other][0]) < tol, other
        assert np.isfinite(out[other][3]).all() and rel_l2(out["small"][3], out[other][3]) < 1e-13, other


@pytest.mark.parametrize("tiles", ["t", "u"])
@pytest.mark.parametrize("shape", [(130, 97), (54, 46), (55, 47), (54, 54), (55, 55), (301, 211)])
def test_strip_kernel_matches_the_per_stage_schedule(gpu, monkeypatch, shape, tiles):
    """The strip kernel (scheme 2 with ODINN_FUSED_TILES=t) against the five per-stage kernels (scheme 1) on
    ragged grids around its 54x46 tile, with a constant A and with a gridded A field: equal to rounding under a
    fixed dt, within the solver tolerance under step-size control (see test_fused_tile_sizes_are_equivalent);
    and against the oracle's integrator.  tiles: "t" = 7 rows per thread (54x46 tiles), "u" = 8 rows (54x54)."""
    sched_env(monkeypatch, FUSED_TILES=tiles)
    nx, ny = shape
    H0, B = O.synthetic_valley(nx, ny, 50.0)
    rng = np.random.default_rng(7)
    Af = np.asfortranarray(4e-17 * (1.0 + 0.5 * rng.random((nx - 1, ny - 1))))
    ts = [2010.0, 2010.2, 2010.4]
    for field in (False, True):
        res = {}
        for scheme in (1, 2):
            b = gpu.GlacierBatch([shape], [50.0], A=[4e-17])
            b.set_fields(0, H0, B)
            if field:
                b.set_A_field(0, Af)
            st = b.solve(ts, reltol=1e-8, scheme=scheme)
            ad = (b.snapshot(0, 2), st[0].naccept + st[0].nreject)
            b.solve(ts[:2], fixed_dt=0.002, scheme=scheme)
            res[scheme] = ad + (b.snapshot(0, 1),)
            b.close()
        assert abs(res[1][1] - res[2][1]) <= 1, (shape, field)
        assert rel_l2(res[2][0], res[1][0]) < 1e-6, (shape, field)  # step counts may differ by one: integration-error level
        assert np.isfinite(res[2][2]).all() and rel_l2(res[2][2], res[1][2]) < 1e-13, (shape, field)
    law = O.Law(kind=O.LAW_CONST_A, A=4e-17)
    b = gpu.GlacierBatch([shape], [50.0], A=[4e-17])
    b.set_fields(0, H0, B)
    b.solve(ts, reltol=1e-8, scheme=2)
    snaps, st, _ = O.forward(O.Glacier(H0, B, 50.0, 50.0, O.Phys()), law, O.SimConfig(tstops=ts, reltol=1e-8))
    assert rel_l2(b.snapshot(0, 2), snaps[2]) < 1e-6  # adaptive run, own step sequence: the north-star tolerance
    b.close()
    # the strip kernel DIRECTLY against the oracle's integrator under a fixed dt (same step sequence on both sides):
    # agreement to rounding, constant A and gridded A
    ph = O.Phys()
    for field in (False, True):
        lawf = O.Law(kind=O.LAW_CONST_A, A=Af if field else 4e-17)
        f = lambda H: O.sia2d_rhs(H, B, 50.0, 50.0, ph, lawf)
        ref, sto, _ = O.solve(f, H0, ts[:2], fixed_dt=0.002)
        b = gpu.GlacierBatch([shape], [50.0], A=[4e-17])
        b.set_fields(0, H0, B)
        if field:
            b.set_A_field(0, Af)
        stg = b.solve(ts[:2], fixed_dt=0.002, scheme=2)
        assert stg[0].naccept == sto.naccept
        assert rel_l2(b.snapshot(0, 1), ref[1]) < 1e-11, (shape, field)
        b.close()


def test_randomised_shapes_and_states(gpu):
    """Seeded sweep over ragged shapes (around the 64x16 tile and the 54x40 / 54x8 fused tiles), rough
    states with negative thickness, ice-free patches, generic exponents: RHS, both H-VJP stencils and
    the theta-VJP against the oracle; one RDPK3Sp35 step sequence against the oracle's."""
    rng = np.random.default_rng(2024)
    shapes = [(3, 3), (4, 7), (17, 65), (63, 17), (64, 16), (65, 33), (54, 40), (55, 41), (108, 8), (109, 9),
              (128, 31), (70, 100), (129, 47), (200, 131)]
    for k, (nx, ny) in enumerate(shapes):
        n = 3.0 if k % 3 else 2.7
        C = 0.0 if k % 2 else 5e-8
        ph = O.Phys(n=n, C=C, p=3.0, q=1.0, eta0=1.0 if k % 4 else 0.8)
        x = np.linspace(0, 1, nx)[:, None]
        y = np.linspace(0, 1, ny)[None, :]
        B = 500.0 + 300.0 * x + 80.0 * np.sin(7 * x + 3 * y) + 5.0 * rng.standard_normal((nx, ny))
        H = 120.0 * np.exp(-((x - 0.5) ** 2 + (y - 0.4) ** 2) / 0.08) + 15.0 * rng.standard_normal((nx, ny)) - 20.0
        if k % 3 == 0:
            H[: nx // 2, :] = 0.0  # an ice-free half (exact-shortcut tiles next to active ones)
        H, B = np.asfortranarray(H), np.asfortranarray(B)
        lam = rng.standard_normal((nx, ny))
        A = 3e-17
        law = O.Law(kind=O.LAW_CONST_A, A=A)
        b = gpu.GlacierBatch([(nx, ny)], [37.0], [53.0], phys=[gpu.PhysicalParameters(**ph.__dict__)], A=[A])
        b.set_fields(0, np.maximum(H, 0.0), B)
        ref = O.sia2d_rhs(H, B, 37.0, 53.0, ph, law)
        tol = 1e-11 if n == 3.0 else 1e-10
        assert rel_l2(b.dhdt(0, H), ref) < tol or not ref.any(), (nx, ny)
        gv = O.vjp_H(lam, H, B, 37.0, 53.0, ph, law)
        assert rel_l2(b.vjp_H(0, lam, H), gv) < 10 * tol or not gv.any(), (nx, ny)
        gt = O.vjp_theta(lam, H, B, 37.0, 53.0, ph, law)
        assert abs(b.vjp_theta(0, lam, H)[0] - gt[0]) <= 10 * tol * max(abs(gt[0]), 1e-300), (nx, ny)
        b.set_vjp_method(gpu._lib.VJP_CONTINUOUS)
        gc = O.vjp_H_continuous(lam, H, B, 37.0, 53.0, ph, law)
        assert rel_l2(b.vjp_H(0, lam, H), gc) < 10 * tol or not gc.any(), (nx, ny)
        b.set_vjp_method(gpu._lib.VJP_DISCRETE)
        if nx >= 17 and ny >= 17:
            H0 = np.maximum(H, 0.0)
            f = lambda u: O.sia2d_rhs(u, B, 37.0, 53.0, ph, law)
            dt = 2e-4
            snaps, _, _ = O.solve(f, H0, [0.0, 4 * dt], fixed_dt=dt)
            b.solve([0.0, 4 * dt], fixed_dt=dt)
            assert rel_l2(b.snapshot(0, 1), snaps[1]) < 1e-11, (nx, ny)
            b.solve([0.0, 4 * dt], fixed_dt=dt, scheme=1)
            assert rel_l2(b.snapshot(0, 1), snaps[1]) < 1e-11, (nx, ny)
        b.close()


def test_strip_kernel_randomised_shapes_fixed_dt(gpu, monkeypatch):
    """Seeded sweep of ragged shapes around the strip kernel's 54x46 output tile (and far from it), rough states
    with ice-free patches and negative thickness: three fixed steps of the strip kernel equal the five per-stage
    kernels to rounding, batched as ONE launch over all glaciers (both strip variants)."""
    rng = np.random.default_rng(99)
    shapes = [(3, 3), (5, 60), (53, 45), (54, 46), (55, 47), (107, 93), (109, 91), (64, 56), (65, 57), (200, 17), (17, 200), (163, 139)]
    fields = []
    for k, (nx, ny) in enumerate(shapes):
        x = np.linspace(0, 1, nx)[:, None]
        y = np.linspace(0, 1, ny)[None, :]
        B = 400.0 + 250.0 * x + 60.0 * np.sin(5 * x + 4 * y) + 3.0 * rng.standard_normal((nx, ny))
        H = 150.0 * np.exp(-((x - 0.45) ** 2 + (y - 0.55) ** 2) / 0.06) + 10.0 * rng.standard_normal((nx, ny)) - 15.0
        if k % 3 == 1:
            H[:, : ny // 2] = 0.0
        fields.append((np.asfortranarray(np.maximum(H, 0.0)), np.asfortranarray(B)))
    res = {}
    for key, scheme, tiles in (("staged", 1, "t"), ("t", 2, "t"), ("u", 2, "u")):
        sched_env(monkeypatch, FUSED_TILES=tiles)
        b = gpu.GlacierBatch(shapes, [40.0] * len(shapes), [55.0] * len(shapes), A=[3e-17] * len(shapes))
        for k, (H, B) in enumerate(fields):
            b.set_fields(k, H, B)
        b.solve([0.0, 3e-4], fixed_dt=1e-4, scheme=scheme, dense=scheme - 1)
        res[key] = [b.snapshot(k, 1) for k in range(len(shapes))]
        b.close()
    for key in ("t", "u"):
        for k, shp in enumerate(shapes):
            assert np.isfinite(res[key][k]).all(), (key, shp)
            assert rel_l2(res[key][k], res["staged"][k]) < 1e-13, (key, shp)


def test_self_controlled_step_loop_matches_the_three_launch_loop(gpu, monkeypatch):
    """ODINN_STEP_SC=1: every workgroup of the strip kernel decides the previous attempt of its glacier itself (error
    norm, PID controller, stop handling) and stores the snapshots -- no k_controller / k_poststep launches.  Same
    kernel arithmetic and the same decisions as the three-launch loop: same step counts, same snapshots, on a ragged
    batch with rejections, many stops and a mass balance on half of the glaciers (applied on load by the
    self-controlled kernel, in place by k_poststep)."""
    sched_env(monkeypatch, FUSED_TILES="t")
    shapes = [(130, 97), (96, 80), (201, 103), (54, 46)]
    ts = [2010.0 + 0.05 * j for j in range(9)]
    out = {}
    # "0": the three-launch loop (k_poststep applies the mass balance in place); "2": the two-launch loop of large batches
    # (step kernel + controller; snapshot and mass balance on load, GState::pad bit 2 kept by the controller); "1": the
    # self-controlled loop
    for sc in ("0", "2", "1"):
        sched_env(monkeypatch, STEP_SC="1" if sc == "1" else "0")
        sched_env(monkeypatch, SNAP_ON_LOAD="0" if sc == "0" else "1")
        b = gpu.GlacierBatch(shapes, [50.0] * 4, A=[8e-17, 4e-17, 6e-17, 2e-17])
        for k, (nx, ny) in enumerate(shapes):
            H0, B = O.synthetic_valley(nx, ny, 50.0)
            b.set_fields(k, H0, B)
            if k % 2 == 0:  # a mass balance with elevation feedback on two of the four glaciers, applied at every stop
                mb = _mb(H0, B)
                b.set_mass_balance(k, mb.mb0, mb.dmb_dS, mb.S_ref, mb.mb_max)
        st = b.solve(ts, mb_times=ts[1:], reltol=1e-6, dt0=0.004)  # a large first step: rejections
        out[sc] = ([[b.snapshot(k, j) for j in range(len(ts))] + [b.H(k)] for k in range(4)],
                   [(s.naccept, s.nreject, s.t_final, s.dt_last) for s in st])
        st2 = b.solve(ts[:3], mb_times=ts[1:3], fixed_dt=0.0025)
        out[sc] += ([b.snapshot(k, 2) for k in range(4)], [(s.naccept, s.nreject) for s in st2])
        b.close()
    a = out["0"]
    assert sum(r for _, r, _, _ in a[1]) > 0
    for key in ("2", "1"):
        f = out[key]
        assert [x[:2] for x in a[1]] == [x[:2] for x in f[1]] and a[3] == f[3], (key, a[1], f[1])
        # With today's compiler the kernel instantiations generate the same arithmetic and everything below is
        # bit-identical (np.array_equal holds); the assertions allow what a different FMA contraction could change.
        for k in range(4):
            for j in range(len(ts) + 1):  # every snapshot and the final state (which carries the last mass balance)
                assert rel_l2(f[0][k][j], a[0][k][j]) < 1e-8 or not a[0][k][j].any(), (key, k, j)
            assert np.isfinite(a[2][k]).all() and rel_l2(f[2][k], a[2][k]) < 1e-13, (key, k)


def test_largest_single_gpu_configuration_64x1024(gpu, monkeypatch):
    """BASELINE configs[4] resident on ONE GPU (64 x 1024^2, the largest single-GPU configuration): size-independent
    properties of the fused step on the full batch -- volume conservation per glacier (flux form, no mass balance, the
    caps stay off the boundary ring) and batch-composition independence (a glacier stepped inside the batch of 64 equals,
    bit for bit, the same glacier stepped alone with the same kernel form)."""
    from bench import make_glacier

    sched_env(monkeypatch, FUSED_TILES="u")  # the 8-row strip kernel for both batch sizes
    sched_env(monkeypatch, STEP_SC="0")      # and the same three-launch loop
    n, G = 1024, 64
    rng = np.random.default_rng(5)
    base = [make_glacier(n, k) for k in range(4)]
    # 64 glaciers from 4 distinct fields x 16 scalings of the thickness (cheap to build, all different)
    scal = [0.5 + 0.03 * (k // 4) for k in range(G)]
    b = gpu.GlacierBatch([(n, n)] * G, [100.0] * G, A=[base[k % 4][2] for k in range(G)])
    for k in range(G):
        b.set_fields(k, base[k % 4][0] * scal[k], base[k % 4][1])
    ts = [0.0, 0.001]
    st = b.solve(ts, fixed_dt=0.00025)
    assert all(s.naccept == 4 for s in st)
    picks = [0, 37, 63]
    got = {k: b.snapshot(k, 1) for k in picks}
    vol_ok = []
    for k in range(0, G, 7):
        H1 = b.snapshot(k, 1) if k not in got else got[k]
        V0 = (base[k % 4][0] * scal[k]).sum()
        vol_ok.append(abs(H1.sum() - V0) <= 1e-12 * V0)
        assert np.isfinite(H1).all() and H1.min() >= 0.0
        assert np.all(H1[0, :] == 0) and np.all(H1[:, -1] == 0)
    assert all(vol_ok)
    b.close()
    for k in picks:
        b1 = gpu.GlacierBatch([(n, n)], [100.0], A=[base[k % 4][2]])
        b1.set_fields(0, base[k % 4][0] * scal[k], base[k % 4][1])
        b1.solve(ts, fixed_dt=0.00025)
        assert np.array_equal(b1.snapshot(0, 1), got[k]), k
        b1.close()


@pytest.mark.parametrize("tiles", ["small", "large"])
@pytest.mark.parametrize("case", ["Y_default", "Y_w16", "Y_light", "Y_gelu5", "Y_wide", "U_default", "U_gelu5"])
def test_inlined_mlp_fused_step_matches_per_stage_and_oracle(gpu, monkeypatch, case, tiles):
    """LawY / LawU with the network INLINED in the temporally fused step kernel (k_rk_fused<LM >= 2>: one launch per
    RDPK3Sp35 step, the MLP evaluated once per dual node and stage inside the stencil) against the five per-stage
    kernels and against the oracle's integrator: equal to rounding under a fixed dt, within the solver tolerance under
    step-size control.  Compile-time architectures (LM 3, 4, 5) and the run-time ones (LM 2 "gelu5", LM 6 "wide"); both tile heights."""
    sched_env(monkeypatch, FUSED_TILES=tiles)
    ph = O.Phys()
    widths, acts = {"default": ([2, 3, 10, 3, 1], [1, 1, 1, 2]), "w16": ([2, 16, 16, 1], [1, 1, 2]),
                    "light": ([2, 3, 1], [1, 2]), "gelu5": ([2, 5, 10, 5, 1], [3, 3, 3, 1]),
                    "wide": ([2, 5, 8, 20, 30, 10, 1], [3, 3, 1, 1, 1, 2])}[case.split("_")[1]]
    isY = case.startswith("Y")
    if isY:
        om, gm, th = _mlp_pair(gpu, widths, acts, [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
        law = O.Law(kind=O.LAW_NN_Y, mlp=om, theta=th, T=-5.0)
        kind = gpu.LAW_NN_Y
    else:
        om, gm, th = _mlp_pair(gpu, widths, acts, [(0.0, 300.0), (0.0, 0.5)], O.POST_EXPMAX, 0.0, 50.0)
        law = O.Law(kind=O.LAW_NN_U, mlp=om, theta=th)
        kind = gpu.LAW_NN_U
    nx, ny = 131, 97  # ragged around the 54 x 40 / 54 x 8 tiles
    H0, B = O.synthetic_icecap(nx, ny, 100.0)
    H0 = H0 * 0.4
    ts = [0.0, 0.02, 0.05] if isY else [0.0, 2e-3, 5e-3]
    fdt = 0.004 if isY else 4e-4
    res = {}
    for scheme in (1, 2):
        b = gpu.GlacierBatch([(nx, ny)], [100.0], phys=[gpu.PhysicalParameters(**ph.__dict__)])
        b.set_fields(0, H0, B)
        b.set_law(kind, gm, th)
        st = b.solve(ts, reltol=1e-8, scheme=scheme)
        ad = (b.snapshot(0, 2), st[0].naccept + st[0].nreject)
        stf = b.solve(ts[:2], fixed_dt=fdt, scheme=scheme)
        res[scheme] = ad + (b.snapshot(0, 1), stf[0].naccept)
        b.close()
    assert np.isfinite(res[2][2]).all() and rel_l2(res[2][2], res[1][2]) < 1e-12, case
    assert abs(res[1][1] - res[2][1]) <= 1 and rel_l2(res[2][0], res[1][0]) < 1e-6, case
    f = lambda H: O.sia2d_rhs(H, B, 100.0, 100.0, ph, law)
    ref, sto, _ = O.solve(f, H0, ts[:2], fixed_dt=fdt)
    assert sto.naccept == res[2][3]
    assert rel_l2(res[2][2], ref[1]) < 1e-10, case  # 5 steps of exp/log-laden arithmetic (device softplus: <= 2 ulp)
    assert np.abs(res[2][2] - H0).max() > 1e-6  # the steps moved the ice


def test_Y_law_theta_gradient_interpolation_modes(gpu):
    """∂Diffusivity∂θ of the Y law (target_D_hybrid.jl:98-166): the reference's default `interpolation = :Linear`
    (gradients on the 2 n_interp_half knots of create_interpolation(H̄), interpolated linearly; on the device: sort by
    H̄, per-interval fixed-order sums, contraction over knots) and `:None` (exact per node), each against the oracle's
    restatement of the same branch; per-glacier knots in a batch; ice-free glacier; the default is :Linear(75)."""
    ph = O.Phys()
    om, gm, th = _mlp_pair(gpu, [2, 3, 10, 3, 1], [1, 1, 1, 2], [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
    shapes = [(80, 48), (131, 97), (40, 33)]
    fields = []
    for k, (nx, ny) in enumerate(shapes):
        H0, B = (O.synthetic_icecap(nx, ny, 100.0) if k != 1 else O.synthetic_valley(nx, ny, 100.0))
        fields.append((H0 * (0.4 if k != 1 else 1.0), B))
    fields[2] = (np.zeros_like(fields[2][0]), fields[2][1])  # no ice at all
    Ts = [-5.0, -2.0, -7.0]
    b = gpu.GlacierBatch(shapes, [100.0] * 3, T=Ts)
    for k, (H0, B) in enumerate(fields):
        b.set_fields(k, H0, B)
    b.set_law(gpu.LAW_NN_Y, gm, th)
    rng = np.random.default_rng(5)
    lams = [rng.standard_normal(s) for s in shapes]
    for mode, n in (("default", 75), ("linear", 20), ("linear", 200), ("none", 0)):
        if mode == "linear":
            b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR, n)
        elif mode == "none":
            b.set_grad_interpolation(gpu._lib.GRAD_INTERP_NONE, 75)
        for k, (H0, B) in enumerate(fields):
            law = O.Law(kind=O.LAW_NN_Y, mlp=om, theta=th, T=Ts[k], interpolation=None if mode == "default" else mode,
                        n_interp_half=None if mode == "default" else n)
            ref = O.vjp_theta(lams[k], H0, B, 100.0, 100.0, ph, law)
            got = b.vjp_theta(k, lams[k], H0)
            if k == 2:
                assert np.all(got == 0.0) and np.all(ref == 0.0)
            else:
                assert rel_l2(got, ref) < 1e-10, (mode, n, k)
    # the two branches differ by the interpolation error only
    b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR, 75)
    gl = b.vjp_theta(0, lams[0], fields[0][0])
    b.set_grad_interpolation(gpu._lib.GRAD_INTERP_NONE, 75)
    gn = b.vjp_theta(0, lams[0], fields[0][0])
    assert 1e-12 < rel_l2(gl, gn) < 5e-3
    b.close()
    # A-type laws have no spatial law gradient: the linear branch is rejected
    b = gpu.GlacierBatch([(40, 33)], [100.0])
    with pytest.raises(gpu.OdinnError):
        b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR, 75)
    b.close()


@pytest.mark.parametrize("arch", ["default", "w16", "runtime"])
def test_Y_law_interpolation_batched_and_per_glacier_sequences_agree(gpu, monkeypatch, arch):
    """The `:Linear` gradient of a whole ragged batch through its three launch schedules: one sequence for all glaciers
    (one radix sort of every dual node by Hbar + one stable sort by glacier, knots and interval sums with a block row per
    glacier, the knot contraction as one wave-reduced backprop; the default), one sequence per glacier on side streams,
    and one per glacier on the batch's stream.  Same knots, same interval sums; only the contraction's rounding differs."""
    ph = O.Phys()
    widths, acts = {"default": ([2, 3, 10, 3, 1], [1, 1, 1, 2]), "w16": ([2, 16, 16, 1], [1, 1, 2]), "runtime": ([2, 5, 4, 1], [4, 1, 2])}[arch]
    om, gm, th = _mlp_pair(gpu, widths, acts, [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
    shapes = [(80, 48), (131, 97), (40, 33), (66, 70)]
    ts = [0.0, 0.5, 1.0]
    out = []
    for batch, streams in (("1", "8"), ("0", "8"), ("0", "1")):
        sched_env(monkeypatch, INTERP_BATCH=batch)
        sched_env(monkeypatch, INTERP_STREAMS=streams)
        b = gpu.GlacierBatch(shapes, [100.0] * 4, T=[-5.0, -2.0, -7.0, -4.0])
        for k, (nx, ny) in enumerate(shapes):
            H0, B = O.synthetic_icecap(nx, ny, 100.0) if k != 1 else O.synthetic_valley(nx, ny, 100.0)
            H0 = H0 * (0.4 if k != 1 else 1.0) * (0.0 if k == 2 else 1.0)  # glacier 2: no ice at all
            b.set_fields(k, H0, B)
            b.set_reference(k, ts, [H0 * (1.0 - 0.05 * j) for j in range(3)], 3)
        b.set_law(gpu.LAW_NN_Y, gm, th)
        L, g = b.loss_grad(ts, theta=th, reltol=1e-8)
        Lc, gc = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
        out.append((L, np.array(g, dtype=float).ravel(), np.array(gc, dtype=float).ravel()))
        b.close()
    for o in out[1:]:
        assert o[0] == out[0][0]
        assert rel_l2(o[1], out[0][1]) < 1e-12 and rel_l2(o[2], out[0][2]) < 1e-12
    assert np.linalg.norm(out[0][1]) > 0.0


@pytest.mark.parametrize("adjoint", ["discrete", "continuous"])
def test_Y_law_interpolation_by_selection_matches_the_sorted_contraction(gpu, monkeypatch, adjoint):
    """The sort-free `:Linear` contraction (k_interp.hip: order statistics by histogram + radix selection, interval sums in
    fixed-point accumulators; the default) against the radix sort of the active nodes it replaces (ODINN_INTERP_SELECT=0):
    the same knots (the selected order statistics ARE the sorted array's entries) and interval sums that differ only in the
    rounding of their additions -- gradients equal to 1e-12; bitwise repeatable although its additions are atomic (integer
    accumulators).  Ragged batch: an ice-free glacier, a valley glacier, and an ice cap whose plateau puts thousands of nearly
    equal thicknesses into one histogram bin (the selection then narrows the bin digit by digit)."""
    ph = O.Phys()
    om, gm, th = _mlp_pair(gpu, [2, 3, 10, 3, 1], [1, 1, 1, 2], [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
    shapes = [(150, 140), (131, 97), (40, 33), (66, 70)]
    ts = [0.0, 0.25, 0.5]
    out = []
    for sel in ("1", "0", "1"):
        monkeypatch.setenv("ODINN_INTERP_SELECT", sel)
        b = gpu.GlacierBatch(shapes, [100.0] * 4, T=[-5.0, -2.0, -7.0, -4.0])
        for k, (nx, ny) in enumerate(shapes):
            H0, B = O.synthetic_icecap(nx, ny, 100.0) if k != 1 else O.synthetic_valley(nx, ny, 100.0)
            if k == 0:  # plateau: thicknesses within 0.3 mm of 250 m on a third of the cap
                H0 = np.asfortranarray(np.minimum(H0, 250.0) + 1e-6 * H0)
            H0 = H0 * (0.4 if k not in (0, 1) else 1.0) * (0.0 if k == 2 else 1.0)
            b.set_fields(k, H0, B)
            b.set_reference(k, ts, [H0 * (1.0 - 0.05 * j) for j in range(3)], 3)
        b.set_law(gpu.LAW_NN_Y, gm, th)
        if adjoint == "discrete":
            L, g = b.loss_grad(ts, theta=th, reltol=1e-8)
        else:
            L, g = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
        out.append((L, np.array(g, dtype=float).ravel()))
        b.close()
    assert out[0][0] == out[1][0] == out[2][0]
    assert np.linalg.norm(out[0][1]) > 0.0
    assert np.array_equal(out[0][1], out[2][1])  # repeatable to the bit
    assert rel_l2(out[0][1], out[1][1]) < 1e-12


def test_U_law_bilinear_contraction_without_a_sort_matches_the_sorted_one(gpu, monkeypatch):
    """Laws.jl:128-169 on the device: the corner sums of the node-grid cells by order-free fixed-point accumulation (64-bit integer
    atomics, two limbs; k_ucell_accum, the default since round 6) against the stable radix sort by grid cell + fixed-order sums it
    replaced (ODINN_INTERP_SELECT=0): 1e-13, and the sort-free form repeats to the bit."""
    om, gm, th = _mlp_pair(gpu, [2, 3, 10, 3, 1], [1, 1, 1, 2], [(0.0, 300.0), (0.0, 0.5)], O.POST_EXPMAX, 0.0, 50.0)
    shapes = [(131, 97), (80, 48)]
    b = gpu.GlacierBatch(shapes, [100.0] * 2)
    fields = []
    for k, (nx, ny) in enumerate(shapes):
        H0, B = (O.synthetic_valley(nx, ny, 100.0) if k == 0 else O.synthetic_icecap(nx, ny, 100.0))
        fields.append((H0 * (90.0 / H0.max()), B))
        b.set_fields(k, *fields[-1])
    b.set_law(gpu.LAW_NN_U, gm, th)
    rng = np.random.default_rng(11)
    lams = [rng.standard_normal(s) for s in shapes]
    for n in (10, 100, 256):
        b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR, n)
        for k in range(2):
            monkeypatch.delenv("ODINN_INTERP_SELECT", raising=False)
            free = b.vjp_theta(k, lams[k], fields[k][0])
            assert np.array_equal(free, b.vjp_theta(k, lams[k], fields[k][0]))
            monkeypatch.setenv("ODINN_INTERP_SELECT", "0")
            srt = b.vjp_theta(k, lams[k], fields[k][0])
            assert np.linalg.norm(free) > 0 and rel_l2(free, srt) < 1e-13, (n, k, rel_l2(free, srt))
    b.close()


def test_U_law_theta_gradient_bilinear_interpolation(gpu):
    """SIA2D_D_target(interpolation = :Linear) (target_D_pure.jl:179-193): gradients of the U law on the fixed
    (2 n_interp_half)^2 node grid of LawU's p_VJP! (Laws.jl:128-169), bilinear in (Hbar, |grad S|).  On the device: dual
    nodes sorted by grid cell, per-cell corner sums in a fixed order, backprop on the grid nodes that carry weight.
    Against the oracle's restatement; ragged batch with an ice-free glacier; repeatable to the bit; the default of the law
    stays :None; a node with Hbar > 100 fails like the reference's BoundsError and the batch stays usable."""
    ph = O.Phys()
    om, gm, th = _mlp_pair(gpu, [2, 3, 10, 3, 1], [1, 1, 1, 2], [(0.0, 300.0), (0.0, 0.5)], O.POST_EXPMAX, 0.0, 50.0)
    shapes = [(80, 48), (131, 97), (40, 33)]
    fields = []
    for k, (nx, ny) in enumerate(shapes):
        H0, B = (O.synthetic_icecap(nx, ny, 100.0) if k != 1 else O.synthetic_valley(nx, ny, 100.0))
        fields.append((H0 * (90.0 / H0.max()), B))
    fields[2] = (np.zeros_like(fields[2][0]), fields[2][1])
    b = gpu.GlacierBatch(shapes, [100.0] * 3)
    for k, (H0, B) in enumerate(fields):
        b.set_fields(k, H0, B)
    b.set_law(gpu.LAW_NN_U, gm, th)
    rng = np.random.default_rng(5)
    lams = [rng.standard_normal(s) for s in shapes]
    exact = [b.vjp_theta(k, lams[k], fields[k][0]) for k in range(3)]  # the law's default: :None
    for k in range(2):
        ref = O.vjp_theta(lams[k], fields[k][0], fields[k][1], 100.0, 100.0, ph, O.Law(kind=O.LAW_NN_U, mlp=om, theta=th))
        assert rel_l2(exact[k], ref) < 1e-10
    for n in (100, 10, 256):
        b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR, n)
        for k, (H0, B) in enumerate(fields):
            law = O.Law(kind=O.LAW_NN_U, mlp=om, theta=th, interpolation="linear", n_interp_half=n)
            ref = O.vjp_theta(lams[k], H0, B, 100.0, 100.0, ph, law)
            got = b.vjp_theta(k, lams[k], H0)
            if k == 2:
                assert np.all(got == 0.0) and np.all(ref == 0.0)
            else:
                assert rel_l2(got, ref) < 1e-10, (n, k)
                assert np.array_equal(got, b.vjp_theta(k, lams[k], H0))
                assert 1e-12 < rel_l2(got, exact[k]) < 0.5  # interpolation error (the slope axis is [0, 100])
    # outside the interpolant: error, then the batch keeps working
    with pytest.raises(gpu.OdinnError, match="BoundsError"):
        b.vjp_theta(0, lams[0], fields[0][0] * 1.5)
    got = b.vjp_theta(0, lams[0], fields[0][0])
    law = O.Law(kind=O.LAW_NN_U, mlp=om, theta=th, interpolation="linear", n_interp_half=256)
    assert rel_l2(got, O.vjp_theta(lams[0], fields[0][0], fields[0][1], 100.0, 100.0, ph, law)) < 1e-10
    b.close()
    # whole gradients through both adjoints with the :Linear branch on both sides
    nx, ny = 56, 40
    H0, B = O.synthetic_alpine(nx, ny, hmax=80.0, slope=0.1)
    ts = [2010.0 + j / 48.0 for j in range(4)]
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    law = O.Law(kind=O.LAW_NN_U, mlp=om, theta=th, interpolation="linear", n_interp_half=100)
    cfg = O.SimConfig(tstops=ts, reltol=1e-8)
    ref, _, _ = O.forward(gl, law, cfg)
    ref = [r * (1.0 + 0.02 * j) for j, r in enumerate(ref)]
    b = gpu.GlacierBatch([(nx, ny)], [50.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_U, gm, th)
    b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR, 100)
    b.set_reference(0, ts, ref, 3)
    Lo, go = O.loss_and_grad(gl, law, cfg, ref, ts)[:2]
    Lg, gg = b.loss_grad(ts, theta=th, reltol=1e-8)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo) and rel_l2(gg, go) < 1e-5
    Lc, gc = O.loss_and_grad_continuous(gl, law, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=8))[:2]
    Lg2, gg2 = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
    assert abs(Lg2 - Lc) <= 1e-6 * abs(Lc) and rel_l2(gg2, gc) < 1e-4
    b.close()


def test_snapshot_on_load_matches_the_post_step_launch(gpu, monkeypatch):
    """Large batches without a mass balance run TWO launches per step (step kernel, controller): the strip kernel stores
    the snapshot of a stop from the state it loads, finished glaciers flush theirs in the next launch.  Bit-identical
    snapshots, step counts and final states to the three-launch loop (ODINN_SNAP_ON_LOAD=0), ragged glaciers that finish
    at different launches, 7- and 8-row tiles."""
    sched_env(monkeypatch, STEP_SC="0")
    shapes = [(130, 97), (54, 46), (201, 103), (70, 57)]
    As = [4e-17, 1e-17, 6e-17, 2e-17]
    ts = [2010.0 + j / 24.0 for j in range(7)]
    for tiles in ("t", "u"):
        sched_env(monkeypatch, FUSED_TILES=tiles)
        out = {}
        for mode in ("1", "0"):
            sched_env(monkeypatch, SNAP_ON_LOAD=mode)
            b = gpu.GlacierBatch(shapes, [50.0] * 4, A=As)
            for k, (nx, ny) in enumerate(shapes):
                b.set_fields(k, *O.synthetic_valley(nx, ny, 50.0))
            st = b.solve(ts, reltol=1e-8)
            out[mode] = ([(s.naccept, s.nreject) for s in st], [[b.snapshot(k, j) for j in range(len(ts))] for k in range(4)],
                         [b.H(k) for k in range(4)])
            b.close()
        assert out["1"][0] == out["0"][0]
        assert len({c for c in out["1"][0]}) > 1  # the glaciers really finish at different launches
        for k in range(4):
            for j in range(len(ts)):
                assert np.array_equal(out["1"][1][k][j], out["0"][1][k][j]), (tiles, k, j)
            assert np.array_equal(out["1"][2][k], out["0"][2][k])
            assert np.abs(out["1"][1][k][-1] - out["1"][1][k][0]).max() > 0


def test_strip_H_vjp_matches_the_tile_kernel(gpu, monkeypatch):
    """k_vjp_H_strip (strip / face layout, 62 x 62 tiles, MODE 0 and the reverse-Euler step MODE 1 with the loss term)
    against k_vjp_H (64 x 16 LDS tiles, node form) on a ragged batch -- an ice-free glacier, a gridded A field and
    glaciers smaller than one tile included; both against the oracle through the whole discrete adjoint."""
    ph = O.Phys()
    shapes = [(130, 97), (54, 46), (201, 103), (70, 57), (33, 40)]
    ts = [2010.0 + j / 24.0 for j in range(4)]
    rng = np.random.default_rng(3)
    fields, refs, Afs = [], [], []
    for k, (nx, ny) in enumerate(shapes):
        H0, B = (O.synthetic_valley(nx, ny, 60.0) if k % 2 else O.synthetic_icecap(nx, ny, 60.0))
        if k == 3:
            H0 = np.zeros_like(H0)
        fields.append((H0, B))
        Afs.append(np.asfortranarray(3e-17 * (1.0 + 0.5 * rng.uniform(size=(nx - 1, ny - 1)))))
    for afield in (False, True):
        res = {}
        for strip in ("1", "0"):
            sched_env(monkeypatch, VJPH_STRIP=strip)
            sched_env(monkeypatch, VJPTH_STRIP=strip)  # k_vjp_theta_strip against k_vjp_theta in the same go
            b = gpu.GlacierBatch(shapes, [60.0] * len(shapes), A=[4e-17] * len(shapes))
            for k, (H0, B) in enumerate(fields):
                b.set_fields(k, H0, B)
                if afield:
                    b.set_A_field(k, Afs[k])
            b.solve(ts, reltol=1e-8)
            if not refs:
                refs = [[b.snapshot(k, j) * (1.0 + 0.03 * j) for j in range(len(ts))] for k in range(len(shapes))]
            for k in range(len(shapes)):
                b.set_reference(k, ts, refs[k], 3)
            L, g = b.loss_grad(ts, reltol=1e-8)
            lam0 = [b.lambda0(k) for k in range(len(shapes))]
            Gf = [b.grad_field(k) for k in range(len(shapes))] if afield else []
            # the reverse ODE: theta-VJP with H formed from two snapshots at the quadrature nodes, in-place accumulation,
            # lambda from the per-glacier ping-pong buffer
            Lc, gc = b.loss_grad_continuous(ts, reltol=1e-8, n_quadrature=6)
            res[strip] = (L, g, lam0, Gf, Lc, gc)
            b.close()
        assert abs(res["1"][0] - res["0"][0]) <= 1e-13 * abs(res["0"][0])
        assert np.allclose(res["1"][1], res["0"][1], rtol=1e-11, atol=0)
        for k in range(len(shapes)):
            assert rel_l2(res["1"][2][k], res["0"][2][k]) < 1e-12 or np.all(res["0"][2][k] == 0), (afield, k)
            if afield and k != 3:
                assert rel_l2(res["1"][3][k], res["0"][3][k]) < 1e-11, k
        assert abs(res["1"][4] - res["0"][4]) <= 1e-12 * abs(res["0"][4])
        # two adaptive reverse solves: the H-VJP behind the initial step size rounds differently in the two layouts, which
        # can shift the step sequence -- agreement to the integration error (see the fused-vs-staged tests; 1e-16 ... 5e-7 with
        # the default forward schedule, 1.6e-6 seen under ODINN_STEP_SC=0 / ODINN_SCHEME=1, whose forward snapshots differ
        # from the default's in the last bits)
        assert np.linalg.norm(res["1"][5] - res["0"][5]) <= 5e-6 * np.linalg.norm(res["0"][5])
    # the strip kernel against the oracle directly (one glacier per batch: the per-glacier entry point runs it too)
    sched_env(monkeypatch, VJPH_STRIP="1")
    H0, B = fields[0]
    b = gpu.GlacierBatch([shapes[0]], [60.0], A=[4e-17])
    b.set_fields(0, H0, B)
    b.set_A_field(0, Afs[0])
    lam = rng.standard_normal(H0.shape)
    H = np.maximum(H0 + rng.standard_normal(H0.shape), 0.0)
    want = O.vjp_H(lam, H, B, 60.0, 60.0, ph, O.Law(kind=O.LAW_CONST_A, A=Afs[0]))
    assert rel_l2(b.vjp_H(0, lam, H), want) < 1e-11
    b.close()
    sched_env(monkeypatch, VJPTH_STRIP="1")
    b = gpu.GlacierBatch([shapes[0]], [60.0], A=[4e-17])
    b.set_fields(0, H0, B)
    wth = O.vjp_theta(lam, H, B, 60.0, 60.0, ph, O.Law(kind=O.LAW_CONST_A, A=4e-17))
    assert abs(b.vjp_theta(0, lam, H)[0] - wth[0]) <= 1e-11 * abs(wth[0])
    b.close()


@pytest.mark.parametrize("adjoint", ["discrete", "continuous"])
def test_lossH_with_logsum(gpu, adjoint):
    """LossH(loss = LogSum(eps)) (Losses.jl:34-49,207-229): log^2((H + eps) / (H_ref + eps)) on the reference mask in both
    adjoints (tile H-VJP kernel MODE 1, k_loss, k_adj_poststep) against the oracle, with a mass balance."""
    nx, ny = 64, 48
    ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref = _inversion_case(gpu, nx, ny, True)
    cfg.h_log_eps = 0.1
    law0 = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th0, T=-2.0)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-2.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_A_SCALAR, gm, th0)
    b.set_reference(0, ts, ref, 3)
    b.set_mass_balance(0, mb.mb0, mb.dmb_dS, mb.S_ref, mb.mb_max)
    b.set_thickness_loss_function(0.1)
    if adjoint == "discrete":
        Lo, go, lam0 = O.loss_and_grad(gl, law0, cfg, ref, ts)
        Lg, gg = b.loss_grad(ts, theta=th0, mb_times=ts[1:], reltol=1e-8)
    else:
        # LogSum puts 1 / (H + eps) into the loss term: lambda is large and sharp on the thin margin cells, and the adaptive
        # reverse solve resolves it to its tolerance only -- the comparison of lambda(t0) runs at tighter reverse tolerances
        Lo, go, lam0, _ = O.loss_and_grad_continuous(gl, law0, cfg, ref, ts,
                                                     O.ContinuousAdjointCfg(n_quadrature=16, reltol=1e-10, abstol=1e-12))
        Lg, gg = b.loss_grad_continuous(ts, theta=th0, mb_times=ts[1:], reltol=1e-8, n_quadrature=16, adj_reltol=1e-10,
                                        adj_abstol=1e-12)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (ratio, angle, relerr)
    assert rel_l2(b.lambda0(0), lam0) < (1e-5 if adjoint == "discrete" else 1e-4)
    b.solve(ts, mb_times=ts[1:], reltol=1e-8)
    assert abs(b.loss()[0] - Lg) <= 1e-8 * abs(Lg)
    # it is not the L2 loss
    b.set_thickness_loss_function(None)
    L2, _ = b.loss_grad(ts, theta=th0, mb_times=ts[1:], reltol=1e-8)
    assert abs(L2 - Lg) > 1e-3 * abs(Lg)
    b.close()
