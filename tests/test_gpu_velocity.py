"""GPU: surface-velocity path -- surface_V, its discrete VJPs and LossV /
LossHV in the discrete adjoint, against the oracle and the reference's FD thresholds
(test_adjoint_surface_V, default [2e-4, 2e-4, 2e-2], test/SIA2D_adjoint.jl:209-216)."""
import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu


def _batch(gpu, nx, ny, ph, law, T=-5.0):
    H0, B = O.synthetic_alpine(nx, ny)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], phys=[gpu.PhysicalParameters(**ph.__dict__)],
                         A=[law.A if law.kind == O.LAW_CONST_A else 1e-18], T=[T])
    b.set_fields(0, H0, B)
    if law.kind != O.LAW_CONST_A:
        m = law.mlp
        b.set_law(law.kind, gpu.MLPSpec(m.widths, m.acts, m.prescale, m.post_kind, m.post_lo, m.post_hi), law.theta)
    return b, H0, B


def _laws(ph):
    rng = np.random.default_rng(1234)
    mlp = O.default_nn(1, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    th = mlp.init_theta(rng) + 0.05 * rng.standard_normal(mlp.n_params)
    return {"constA": O.Law(kind=O.LAW_CONST_A, A=2.21e-17),
            "nnA": O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th, T=-5.0)}


@pytest.mark.parametrize("lawname", ["constA", "nnA"])
@pytest.mark.parametrize("phys", [O.Phys(), O.Phys(n=3.2, C=7e-8, q=1.0)])
def test_surface_V_and_vjps_match_oracle(gpu, lawname, phys):
    law = _laws(phys)[lawname]
    nx, ny = 70, 53
    b, H0, B = _batch(gpu, nx, ny, phys, law)
    vx, vy = O.surface_V(H0, B, 50.0, 50.0, phys, law)
    Vx, Vy = b.surface_V(0, H0)
    assert rel_l2(Vx[:-1, :-1], vx) < 1e-12 and rel_l2(Vy[:-1, :-1], vy) < 1e-12
    assert np.all(Vx[-1, :] == 0) and np.all(Vx[:, -1] == 0) and np.all(Vy[-1, :] == 0)  # inn1 pairing
    rng = np.random.default_rng(7)
    w1, w2 = rng.standard_normal((nx, ny)), rng.standard_normal((nx, ny))
    assert rel_l2(b.surface_V_vjp_H(0, w1, w2, H0), O.vjp_surface_V_H(w1, w2, H0, B, 50.0, 50.0, phys, law)) < 1e-11
    assert rel_l2(b.surface_V_vjp_theta(0, w1, w2, H0), O.vjp_surface_V_theta(w1, w2, H0, B, 50.0, 50.0, phys, law)) < 1e-11
    b.close()


def test_surface_V_vjp_meets_reference_fd_thresholds(gpu):
    ph = O.Phys(maxA=8e-18)  # test/SIA2D_adjoint.jl:244-246
    mlp = O.default_nn(1, light=True, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    law = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=mlp.init_theta(np.random.default_rng(1234)), T=-5.0)
    nx, ny = 40, 33
    b, H0, B = _batch(gpu, nx, ny, ph, law)
    rng = np.random.default_rng(1234)
    w1, w2 = rng.standard_normal((nx, ny)), rng.standard_normal((nx, ny))

    def f(H):
        Vx, Vy = b.surface_V(0, H)
        return np.sum(Vx * w1) + np.sum(Vy * w2)  # == <Vx, inn1(w1)> + <Vy, inn1(w2)>

    g = b.surface_V_vjp_H(0, w1, w2, H0)
    best = [np.inf] * 3
    f0 = f(H0)
    for eps in (1e-3, 1e-5, 1e-7):
        gn = np.zeros_like(H0)
        for i in range(nx):
            for j in range(ny):
                Hp = H0.copy()
                Hp[i, j] += eps
                gn[i, j] = (f(Hp) - f0) / eps
        best = [min(a, abs(s)) for a, s in zip(best, stats_err_arrays(g, gn))]
    assert best[0] < 2e-4 and best[1] < 2e-4 and best[2] < 2e-2, best
    b.close()


def _velocity_case(nx, ny, ph, k=9, step=1.0 / 96.0):
    H0, B = O.synthetic_alpine(nx, ny, hmax=160.0, slope=0.1)
    ts = [2010.0 + j * step for j in range(k)]
    mlp = O.default_nn(1, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    th_true = mlp.init_theta(np.random.default_rng(42))
    th0 = mlp.init_theta(np.random.default_rng(1234))
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    cfg = O.SimConfig(tstops=ts, reltol=1e-10)
    law_t = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th_true, T=-3.0)
    ref, _, _ = O.forward(gl, law_t, cfg)
    tV = ts[2::2]
    Vref = []
    for t in tV:
        Vx, Vy, V = O.V_from_H(ref[ts.index(t)], B, 50.0, 50.0, ph, law_t)
        # data on the full grid, including (non-zero) values on the last row / column
        Vx[-1, :] = 0.3 * Vx[-2, :]
        Vy[:, -1] = 0.3 * Vy[:, -2]
        Vref.append((np.sqrt(Vx ** 2 + Vy ** 2), Vx, Vy))
    return H0, B, ts, mlp, th_true, th0, gl, cfg, ref, tV, Vref


@pytest.mark.parametrize("kind,component,scale", [("V", "xy", True), ("V", "abs", False), ("HV", "xy", True),
                                                  ("V", "log", False), ("HV", "log", True)])
def test_loss_grad_with_velocity_losses(gpu, kind, component, scale):
    """odinn_loss_grad with LossV / LossHV == the oracle's restatement of gradient.jl:129-275 with
    backward_loss(::LossV) (Losses.jl:338-390) and LossHV (Losses.jl:395-440)."""
    ph = O.Phys()
    nx, ny = 64, 48
    H0, B, ts, mlp, th_true, th0, gl, cfg, ref, tV, Vref = _velocity_case(nx, ny, ph)
    log_eps = 0.1 if component == "log" else None  # LossV(loss = LogSum(), component = :abs) (runtests.jl:165-167)
    component = "abs" if component == "log" else component
    vspec = O.LossVSpec(component=component, scale_loss=scale, log_eps=log_eps)
    law0 = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th0, T=-3.0)
    Lo, go, lam0 = O.loss_and_grad_HV(gl, law0, cfg, ref, ts, Vref, tV, vspec, loss_kind=kind, scaling=2.5)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-3.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_A_SCALAR, gpu.MLPSpec(mlp.widths, mlp.acts, None, O.POST_AFFINE, ph.minA, ph.maxA), th0)
    b.set_reference(0, ts, ref, 3)
    b.set_velocity_reference(0, tV, [v[0] for v in Vref], [v[1] for v in Vref], [v[2] for v in Vref])
    b.set_loss({"V": gpu._lib.LOSS_V, "HV": gpu._lib.LOSS_HV}[kind], component, scale, 2.5)
    b.set_velocity_loss_function(log_eps)
    Lg, gg = b.loss_grad(ts, theta=th0, reltol=1e-10)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo), (Lg, Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (ratio, angle, relerr)
    assert rel_l2(b.lambda0(0), lam0) < 1e-5
    assert abs(b.loss()[0] - Lg) <= 1e-8 * abs(Lg)  # forward-only loss == loss from the reverse sweep
    # end-to-end FD of the GPU forward loss (reference thresholds for LossV runs, runtests.jl:153-170: [1e-2,...])
    def loss_at(th):
        b.set_theta(th)
        b.solve(ts, reltol=1e-10)
        return float(b.loss()[0])

    idx = np.arange(0, gg.size, 9)
    gn = np.zeros_like(gg)
    for q in idx:
        e = np.zeros_like(gg)
        e[q] = 1e-4
        gn[q] = (loss_at(th0 + e) - loss_at(th0 - e)) / 2e-4
    ratio, angle, relerr = stats_err_arrays(gg[idx], gn[idx])
    assert abs(ratio) < 1e-2 and abs(angle) < 1e-7 and relerr < 1e-2, (ratio, angle, relerr)
    b.close()


def _u_law(gpu, ph, arch="default"):
    from test_gpu_parity import _mlp_pair
    widths, acts = {"default": ([2, 3, 10, 3, 1], [1, 1, 1, 2]), "custom": ([2, 5, 10, 5, 1], [3, 3, 3, 1])}[arch]
    om, gm, th = _mlp_pair(gpu, widths, acts, [(0.0, 300.0), (0.0, 0.5)], O.POST_EXPMAX, 0.0, 50.0)
    return om, gm, th


@pytest.mark.parametrize("arch", ["default", "custom"])
def test_surface_V_with_the_U_law_target_D(gpu, arch):
    """Target :D (target_D_pure.jl:206-255): Velocity^ = U / f, its partials by central differences of the law (1e-4 in Hbar,
    1e-6 in |grad S|), dVelocity^/dtheta by per-node backprop; surface_V and both VJPs against the oracle (the H-VJP to the
    agreement of two finite-difference evaluations in different summation orders)."""
    ph = O.Phys()
    om, gm, th = _u_law(gpu, ph, arch)
    law = O.Law(kind=O.LAW_NN_U, mlp=om, theta=th, fV=0.8)  # f_surface_velocity_factor of the reference's test (test_grad_loss.jl:120)
    nx, ny = 70, 53
    H0, B = O.synthetic_alpine(nx, ny)
    b = gpu.GlacierBatch([(nx, ny)], [50.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_U, gm, th)
    b.set_surface_velocity_factor(0.8)
    vx, vy = O.surface_V(H0, B, 50.0, 50.0, ph, law)
    Vx, Vy = b.surface_V(0, H0)
    assert rel_l2(Vx[:-1, :-1], vx) < 1e-12 and rel_l2(Vy[:-1, :-1], vy) < 1e-12
    rng = np.random.default_rng(7)
    w1, w2 = rng.standard_normal((nx, ny)), rng.standard_normal((nx, ny))
    assert rel_l2(b.surface_V_vjp_H(0, w1, w2, H0), O.vjp_surface_V_H(w1, w2, H0, B, 50.0, 50.0, ph, law)) < 1e-6
    assert rel_l2(b.surface_V_vjp_theta(0, w1, w2, H0), O.vjp_surface_V_theta(w1, w2, H0, B, 50.0, 50.0, ph, law)) < 1e-10
    b.close()


@pytest.mark.parametrize("adjoint", ["discrete", "continuous"])
def test_lossV_with_the_U_law_target_D(gpu, adjoint):
    """LossV with target :D -- the reference's 'continuous adjoint with discrete VJP (loss V)', runtests.jl:192-194 and
    :201-203 (custom NN) -- through both adjoints against the oracle, two glaciers with different velocity dates."""
    ph = O.Phys()
    om, gm, th = _u_law(gpu, ph)
    law = O.Law(kind=O.LAW_NN_U, mlp=om, theta=th, fV=0.8)
    step = 1.0 / 96.0
    ts = [2010.0 + j * step for j in range(7)]
    shapes = [(56, 40), (64, 48)]
    b = gpu.GlacierBatch(shapes, [50.0] * 2)
    vspec = O.LossVSpec(component="xy", scale_loss=True)
    Lo, go = 0.0, 0.0
    for k, (nx, ny) in enumerate(shapes):
        H0, B = O.synthetic_alpine(nx, ny, hmax=150.0, slope=0.1)
        gl = O.Glacier(H0, B, 50.0, 50.0, ph)
        cfg = O.SimConfig(tstops=ts, reltol=1e-10)
        ref, _, _ = O.forward(gl, law, cfg)
        tV = ts if adjoint == "continuous" else ts[1 + k::2]  # the continuous adjoint interpolates the maps over tspan
        Vref = []
        for t in tV:
            Vx, Vy, V = O.V_from_H(ref[ts.index(t)], B, 50.0, 50.0, ph, law)
            Vref.append((1.1 * V, 1.1 * Vx, 1.1 * Vy))
        b.set_fields(k, H0, B)
        b.set_reference(k, ts, ref, 3)
        b.set_velocity_reference(k, tV, [v[0] for v in Vref], [v[1] for v in Vref], [v[2] for v in Vref])
        if adjoint == "discrete":
            l, g, _ = O.loss_and_grad_HV(gl, law, cfg, ref, ts, Vref, tV, vspec, loss_kind="V")
        else:
            l, g, _, _ = O.loss_and_grad_continuous(gl, law, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=8),
                                                    V_ref=Vref, tV_ref=tV, vspec=vspec, loss_kind="V")
        Lo, go = Lo + l, go + g
    b.set_law(gpu.LAW_NN_U, gm, th)
    b.set_surface_velocity_factor(0.8)
    b.set_loss(gpu._lib.LOSS_V, "xy", True, 1.0)
    if adjoint == "discrete":
        Lg, gg = b.loss_grad(ts, theta=th, reltol=1e-10)
    else:
        Lg, gg = b.loss_grad_continuous(ts, theta=th, reltol=1e-10, n_quadrature=8)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo), (Lg, Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-8 and relerr < 1e-5, (ratio, angle, relerr)
    b.close()


@pytest.mark.parametrize("adjoint", ["seam", "discrete", "continuous"])
def test_velocity_theta_path_of_the_U_law_honours_linear_interpolation(gpu, adjoint):
    """dVelocity^/dtheta of target :D is dU/dtheta / f (target_D_pure.jl:247-255), and dU/dtheta honours target.interpolation
    (:163-193): with `:Linear` the law gradient comes from LawU's (2 n_interp_half)^2 node grid, bilinear in (Hbar, |grad S|)
    -- in the surface-velocity pull-back too, not only in the diffusivity's theta-VJP.  Seam and LossV through both adjoints."""
    ph = O.Phys()
    om, gm, th = _u_law(gpu, ph)
    n_half = 40
    law = O.Law(kind=O.LAW_NN_U, mlp=om, theta=th, fV=0.8, interpolation="linear", n_interp_half=n_half)
    law_none = O.Law(kind=O.LAW_NN_U, mlp=om, theta=th, fV=0.8)
    if adjoint == "seam":
        nx, ny = 70, 53
        H0, B = O.synthetic_alpine(nx, ny, hmax=90.0)
        b = gpu.GlacierBatch([(nx, ny)], [50.0])
        b.set_fields(0, H0, B)
        b.set_law(gpu.LAW_NN_U, gm, th)
        b.set_surface_velocity_factor(0.8)
        rng = np.random.default_rng(11)
        w1, w2 = rng.standard_normal((nx, ny)), rng.standard_normal((nx, ny))
        exact = b.surface_V_vjp_theta(0, w1, w2, H0)
        assert rel_l2(exact, O.vjp_surface_V_theta(w1, w2, H0, B, 50.0, 50.0, ph, law_none)) < 1e-10
        b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR, n_half)
        got = b.surface_V_vjp_theta(0, w1, w2, H0)
        assert rel_l2(got, O.vjp_surface_V_theta(w1, w2, H0, B, 50.0, 50.0, ph, law)) < 1e-10
        assert np.array_equal(got, b.surface_V_vjp_theta(0, w1, w2, H0))
        assert 1e-12 < rel_l2(got, exact) < 0.9  # the two branches differ by the interpolation error
        with pytest.raises(gpu.OdinnError, match="BoundsError"):  # Hbar > 100: the interpolant does not extrapolate
            b.surface_V_vjp_theta(0, w1, w2, H0 * 1.5)
        b.close()
        return
    step = 1.0 / 96.0
    ts = [2010.0 + j * step for j in range(5)]
    nx, ny = 56, 40
    H0, B = O.synthetic_alpine(nx, ny, hmax=80.0, slope=0.1)
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    cfg = O.SimConfig(tstops=ts, reltol=1e-10)
    ref, _, _ = O.forward(gl, law, cfg)
    tV = ts if adjoint == "continuous" else ts[1::2]
    Vref = []
    for t in tV:
        Vx, Vy, V = O.V_from_H(ref[ts.index(t)], B, 50.0, 50.0, ph, law)
        Vref.append((1.1 * V, 1.1 * Vx, 1.1 * Vy))
    vspec = O.LossVSpec(component="xy", scale_loss=True)
    b = gpu.GlacierBatch([(nx, ny)], [50.0])
    b.set_fields(0, H0, B)
    b.set_reference(0, ts, ref, 3)
    b.set_velocity_reference(0, tV, [v[0] for v in Vref], [v[1] for v in Vref], [v[2] for v in Vref])
    b.set_law(gpu.LAW_NN_U, gm, th)
    b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR, n_half)
    b.set_surface_velocity_factor(0.8)
    b.set_loss(gpu._lib.LOSS_V, "xy", True, 1.0)
    if adjoint == "discrete":
        Lo, go, _ = O.loss_and_grad_HV(gl, law, cfg, ref, ts, Vref, tV, vspec, loss_kind="V")
        Lg, gg = b.loss_grad(ts, theta=th, reltol=1e-10)
    else:
        Lo, go, _, _ = O.loss_and_grad_continuous(gl, law, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=8),
                                                  V_ref=Vref, tV_ref=tV, vspec=vspec, loss_kind="V")
        Lg, gg = b.loss_grad_continuous(ts, theta=th, reltol=1e-10, n_quadrature=8)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo), (Lg, Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-8 and relerr < 1e-5, (ratio, angle, relerr)
    b.close()


def _y_law(gpu, ph, arch="default"):
    from test_gpu_parity import _mlp_pair
    widths, acts = {"default": ([2, 3, 10, 3, 1], [1, 1, 1, 2]), "light": ([2, 3, 1], [1, 2]),
                    "wide": ([2, 5, 8, 20, 30, 10, 1], [3, 3, 1, 1, 1, 2])}[arch]  # wide: P = 1194, the reference's diffusivity MWE
    return _mlp_pair(gpu, widths, acts, [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)


@pytest.mark.parametrize("interp", ["linear", "none"])
@pytest.mark.parametrize("C", [0.0, 7e-8])
def test_surface_V_with_the_Y_law_target_D_hybrid(gpu, interp, C):
    """Target :D_hybrid (target_D_hybrid.jl:210-372) AS WRITTEN: Velocity^ with the diffusivity's Gamma, its H-partial with a
    forward difference of compute_D, its slope partial and the theta-weight with Gamma^ (the oracle records the
    inconsistencies; upstream never runs this path).  surface_V and both VJPs against the oracle, the theta-VJP through the
    target's default `:Linear` interpolation of dY/dtheta (knots of create_interpolation) and through `:None`."""
    ph = O.Phys(C=C, p=3.0, q=1.0) if C else O.Phys()
    om, gm, th = _y_law(gpu, ph)
    law = O.Law(kind=O.LAW_NN_Y, mlp=om, theta=th, T=-5.0, interpolation=interp, n_interp_half=20)
    nx, ny = 70, 53
    H0, B = O.synthetic_alpine(nx, ny)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], phys=[gpu.PhysicalParameters(C=C, p=3.0, q=1.0)] if C else None, T=[-5.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_Y, gm, th)
    if interp == "linear":
        b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR, 20)
    else:
        b.set_grad_interpolation(gpu._lib.GRAD_INTERP_NONE, 75)
    vx, vy = O.surface_V(H0, B, 50.0, 50.0, ph, law)
    Vx, Vy = b.surface_V(0, H0)
    assert rel_l2(Vx[:-1, :-1], vx) < 1e-12 and rel_l2(Vy[:-1, :-1], vy) < 1e-12
    rng = np.random.default_rng(7)
    w1, w2 = rng.standard_normal((nx, ny)), rng.standard_normal((nx, ny))
    # (the H-partial contains a forward difference with step 1e-4: two evaluations agree to ~1e-16 / 1e-4 of the law's scale)
    assert rel_l2(b.surface_V_vjp_H(0, w1, w2, H0), O.vjp_surface_V_H(w1, w2, H0, B, 50.0, 50.0, ph, law)) < 1e-7
    assert rel_l2(b.surface_V_vjp_theta(0, w1, w2, H0), O.vjp_surface_V_theta(w1, w2, H0, B, 50.0, 50.0, ph, law)) < 1e-9
    b.close()


def test_velocity_pullback_theta_with_a_wide_network_and_exact_backprop(gpu):
    """interpolation = :None with a network of 1194 parameters (2-5-8-20-30-10-1, scripts/MWEs/inversion_diffusivity): the NW x P
    wave accumulators of k_node_backprop do not fit the LDS -- the one-wavefront-per-workgroup instantiation takes over (the call
    used to fail with ODINN_ERR_UNSUPPORTED).  surface_V's theta-VJP against the oracle."""
    ph = O.Phys()
    om, gm, th = _y_law(gpu, ph, "wide")
    assert th.size == 1194
    law = O.Law(kind=O.LAW_NN_Y, mlp=om, theta=th, T=-5.0, interpolation="none", n_interp_half=20)
    nx, ny = 70, 53
    H0, B = O.synthetic_alpine(nx, ny)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-5.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_Y, gm, th)
    b.set_grad_interpolation(gpu._lib.GRAD_INTERP_NONE, 75)
    rng = np.random.default_rng(7)
    w1, w2 = rng.standard_normal((nx, ny)), rng.standard_normal((nx, ny))
    assert rel_l2(b.surface_V_vjp_theta(0, w1, w2, H0), O.vjp_surface_V_theta(w1, w2, H0, B, 50.0, 50.0, ph, law)) < 1e-9
    b.close()


@pytest.mark.parametrize("adjoint", ["discrete", "continuous"])
@pytest.mark.parametrize("kind", ["V", "HV"])
def test_velocity_losses_with_the_Y_law_target_D_hybrid(gpu, adjoint, kind):
    """LossV / LossHV with target :D_hybrid through both adjoints against the oracle (two glaciers, `:Linear` law gradients)."""
    ph = O.Phys()
    om, gm, th = _y_law(gpu, ph)
    step = 1.0 / 96.0
    ts = [2010.0 + j * step for j in range(5)]
    shapes = [(56, 40), (64, 48)]
    b = gpu.GlacierBatch(shapes, [50.0] * 2, T=[-5.0, -8.0])
    vspec = O.LossVSpec(component="xy", scale_loss=True)
    Lo, go = 0.0, 0.0
    for k, (nx, ny) in enumerate(shapes):
        law = O.Law(kind=O.LAW_NN_Y, mlp=om, theta=th, T=[-5.0, -8.0][k], interpolation="linear", n_interp_half=20)
        H0, B = O.synthetic_alpine(nx, ny, hmax=150.0, slope=0.1)
        gl = O.Glacier(H0, B, 50.0, 50.0, ph)
        cfg = O.SimConfig(tstops=ts, reltol=1e-10)
        ref, _, _ = O.forward(gl, law, cfg)
        tV = ts if adjoint == "continuous" else ts[1 + k::2]
        Vref = []
        for t in tV:
            Vx, Vy, V = O.V_from_H(ref[ts.index(t)], B, 50.0, 50.0, ph, law)
            Vref.append((1.1 * V, 1.1 * Vx, 1.1 * Vy))
        ref = [r * (1.0 - 0.01 * j) for j, r in enumerate(ref)]
        b.set_fields(k, H0, B)
        b.set_reference(k, ts, ref, 3)
        b.set_velocity_reference(k, tV, [v[0] for v in Vref], [v[1] for v in Vref], [v[2] for v in Vref])
        if adjoint == "discrete":
            l, g, _ = O.loss_and_grad_HV(gl, law, cfg, ref, ts, Vref, tV, vspec, loss_kind=kind, scaling=0.7)
        else:
            l, g, _, _ = O.loss_and_grad_continuous(gl, law, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=6),
                                                    V_ref=Vref, tV_ref=tV, vspec=vspec, loss_kind=kind, scaling=0.7)
        Lo, go = Lo + l, go + g
    b.set_law(gpu.LAW_NN_Y, gm, th)
    b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR, 20)
    b.set_loss(gpu._lib.LOSS_V if kind == "V" else gpu._lib.LOSS_HV, "xy", True, 0.7)
    if adjoint == "discrete":
        Lg, gg = b.loss_grad(ts, theta=th, reltol=1e-10)
    else:
        Lg, gg = b.loss_grad_continuous(ts, theta=th, reltol=1e-10, n_quadrature=6)
    b.close()
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo), (Lg, Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-8 and relerr < 1e-5, (ratio, angle, relerr)


@pytest.mark.parametrize("lawkind", ["U", "Y"])
@pytest.mark.parametrize("term", ["avgv", "vreg"])
def test_time_aggregated_velocity_terms_with_the_U_law(gpu, term, lawkind):
    """LossAvgV and VelocityRegularization are generic over the targets in the reference (TimeAggregatedLosses.jl:115-258,
    Regularization.jl:192-245: they go through V_from_H and VJP_lambda_dsurface_V/d{H, theta}): with the U law (target :D),
    next to LossH, through both adjoints against the oracle."""
    from test_gpu_avgv import _sample
    ph = O.Phys()
    if lawkind == "U":
        om, gm, th = _u_law(gpu, ph)
        law = O.Law(kind=O.LAW_NN_U, mlp=om, theta=th, fV=0.8)
    else:  # target :D_hybrid, `:Linear` law gradients (the target's default)
        om, gm, th = _y_law(gpu, ph)
        law = O.Law(kind=O.LAW_NN_Y, mlp=om, theta=th, T=-5.0, interpolation="linear", n_interp_half=20)
    step = 1.0 / 96.0
    ts = [2010.0 + j * step for j in range(7)]
    nx, ny = 56, 40
    H0, B = O.synthetic_alpine(nx, ny, hmax=150.0, slope=0.1)
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    cfg = O.SimConfig(tstops=ts, reltol=1e-10)
    ref, _, _ = O.forward(gl, law, cfg)
    ref = [r * (1.0 - 0.01 * j) for j, r in enumerate(ref)]
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-5.0])
    b.set_fields(0, H0, B)
    b.set_reference(0, ts, ref, 3)
    if lawkind == "U":
        b.set_law(gpu.LAW_NN_U, gm, th)
        b.set_surface_velocity_factor(0.8)
    else:
        b.set_law(gpu.LAW_NN_Y, gm, th)
        b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR, 20)
    if term == "avgv":
        a = _sample(gl, law, cfg, ts, 1, 5, "xy")
        cfg.avgv, cfg.avgv_weight = a, 2.0
        b.set_avgv_reference(0, a.t1, a.t2, a.Vabs, a.Vx, a.Vy)
        b.set_avgv_loss(2.0, a.step, "xy")
    else:
        tV = ts[1::2]
        cfg.vreg_times, cfg.vreg_distance, cfg.vreg_weight = tV, 3, 1e3
        z = [np.zeros((nx, ny))] * len(tV)
        b.set_velocity_reference(0, tV, z, z, z)  # only the dates matter
        b.set_velocity_regularization(1e3, 3)
    Lo, go, _ = O.loss_and_grad(gl, law, cfg, ref, ts)
    Lg, gg = b.loss_grad(ts, theta=th, reltol=1e-10)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo), (Lg, Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-8 and relerr < 1e-5, (term, ratio, angle, relerr)
    Lo, go, _, _ = O.loss_and_grad_continuous(gl, law, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=8))
    Lg, gg = b.loss_grad_continuous(ts, theta=th, reltol=1e-10, n_quadrature=8)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo), (Lg, Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-8 and relerr < 1e-5, (term, ratio, angle, relerr)
    # the term is in there
    L0, g0 = (b.set_avgv_loss(0.0, step, "xy") if term == "avgv" else b.set_velocity_regularization(0.0, 3)) or b.loss_grad(ts, theta=th, reltol=1e-10)
    assert abs(L0 - Lg) > 1e-6 * abs(Lg)
    b.close()
