"""GPU: continuous adjoint -- odinn_loss_grad_continuous ==
SIA2D_grad_batch! with ContinuousAdjoint(VJP_method = DiscreteVJP()) (gradient.jl:276-539)
against the oracle's restatement on the same inputs, and against central finite differences of
the GPU forward loss with the reference's own thresholds [1e-3, 1e-8, 1e-3] (runtests.jl:127)."""
import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays, sched_env
from oracle import sia2d_oracle as O
from test_gpu_parity import _inversion_case, _mb

pytestmark = pytest.mark.gpu


def _batch(gpu, nx, ny, H0, B, gm, th0, ts, ref, mb=None):
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-2.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_A_SCALAR, gm, th0)
    b.set_reference(0, ts, ref, 3)
    if mb is not None:
        b.set_mass_balance(0, mb.mb0, mb.dmb_dS, mb.S_ref, mb.mb_max)
    return b


@pytest.mark.parametrize("use_mb", [False, True])
def test_continuous_adjoint_matches_oracle(gpu, use_mb):
    nx, ny = 64, 48
    ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref = _inversion_case(gpu, nx, ny, use_mb)
    law0 = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th0, T=-2.0)
    adj = O.ContinuousAdjointCfg(n_quadrature=24)
    Lo, go, lam0, st_o = O.loss_and_grad_continuous(gl, law0, cfg, ref, ts, adj)
    b = _batch(gpu, nx, ny, H0, B, gm, th0, ts, ref, mb)
    Lg, gg = b.loss_grad_continuous(ts, theta=th0, mb_times=ts[1:] if use_mb else (), reltol=1e-8, n_quadrature=24)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    # both sides integrate the same reverse ODE adaptively at reltol = abstol = 1e-8
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (ratio, angle, relerr)
    assert rel_l2(b.lambda0(0), lam0) < 1e-5
    sr = b.last_stats_rev[0]
    assert abs(sr.t_final - ts[0]) < 1e-12
    # same controller: the step counts agree (up to a step when an error estimate sits on the accept edge)
    assert abs(sr.naccept - st_o.naccept) <= 2 and abs(sr.nreject - st_o.nreject) <= 2, (sr, st_o)
    b.close()


def test_continuous_adjoint_vs_finite_differences(gpu):
    """test_grad_finite_diff(ContinuousAdjoint(VJP_method = DiscreteVJP()); thres = [1e-3, 1e-8, 1e-3])
    (runtests.jl:127) with monthly snapshots -- the setting where the discrete adjoint of
    gradient.jl:191-253 is unstable on this fast valley (test_gpu_parity.py)."""
    nx, ny = 64, 48
    ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref = _inversion_case(gpu, nx, ny, False, k=7, step=1.0 / 12.0)
    b = _batch(gpu, nx, ny, H0, B, gm, th0, ts, ref)
    L0, g = b.loss_grad_continuous(ts, theta=th0, reltol=1e-10, n_quadrature=60)

    def loss_at(th):
        b.set_theta(th)
        b.solve(ts, reltol=1e-10)
        return b.loss()[0]

    gn = np.zeros_like(g)
    idx = np.arange(0, g.size, 5)
    for q in idx:
        e = np.zeros_like(g)
        e[q] = 1e-4
        gn[q] = (loss_at(th0 + e) - loss_at(th0 - e)) / (2 * e[q])
    ratio, angle, relerr = stats_err_arrays(g[idx], gn[idx])
    assert abs(ratio) < 1e-3 and abs(angle) < 1e-8 and relerr < 1e-3, (ratio, angle, relerr)
    b.close()


def test_continuous_adjoint_ragged_batch_and_other_laws(gpu):
    """Per-glacier reverse solves in one batch (different sizes => different step sequences) equal
    the single-glacier results; constant-A and gridded-A accumulators are fed by the same quadrature."""
    shapes = [(64, 48), (40, 33), (96, 20)]
    ph = O.Phys()
    ts = [2010.0 + j / 24.0 for j in range(5)]
    cases = []
    for k, (nx, ny) in enumerate(shapes):
        H0, B = O.synthetic_valley(nx, ny, 50.0)
        gl = O.Glacier(H0, B, 50.0, 50.0, ph)
        cfg = O.SimConfig(tstops=ts, reltol=1e-8)
        ref, _, _ = O.forward(gl, O.Law(kind=O.LAW_CONST_A, A=3e-17), cfg)
        cases.append((H0, B, gl, cfg, ref))
    A0 = 1.5e-17
    b = gpu.GlacierBatch(shapes, [50.0] * 3, A=[A0] * 3)
    tot_L, tot_g = 0.0, 0.0
    for k, (H0, B, gl, cfg, ref) in enumerate(cases):
        b.set_fields(k, H0, B)
        b.set_reference(k, ts, ref, 3)
        Lo, go, _, _ = O.loss_and_grad_continuous(gl, O.Law(kind=O.LAW_CONST_A, A=A0), cfg, ref, ts,
                                                  O.ContinuousAdjointCfg(n_quadrature=16))
        tot_L += Lo
        tot_g += go[0]
    Lg, gg = b.loss_grad_continuous(ts, reltol=1e-8, n_quadrature=16)
    assert abs(Lg - tot_L) <= 1e-6 * tot_L
    assert abs(gg[0] - tot_g) <= 1e-5 * abs(tot_g)
    b.close()


@pytest.mark.parametrize("onepass", ["1", "0"])
@pytest.mark.parametrize("kind,component,scale", [("V", "xy", True), ("V", "abs", False), ("HV", "xy", True), ("V", "log", True)])
def test_continuous_adjoint_with_velocity_losses(gpu, monkeypatch, kind, component, scale, onepass):
    """ContinuousAdjoint with LossV / LossHV (gradient.jl:291-301, 331-365, 475-503): the velocity term
    enters lambda at the velocity-data snapshots; its explicit theta-dependence is integrated by the
    quadrature with the reference velocities interpolated linearly in time.  Against the oracle, through the one-pass
    node kernel (k_surfV_theta_node, the default for closed-form laws without a dual-grid accumulator) and through the
    interpolate / scale / pull-back / reduce sequence it replaces (ODINN_VQ_ONEPASS=0)."""
    sched_env(monkeypatch, VQ_ONEPASS=onepass)
    from test_gpu_velocity import _velocity_case

    ph = O.Phys()
    nx, ny = 64, 48
    H0, B, ts, mlp, th_true, th0, gl, cfg, ref, tV, Vref = _velocity_case(nx, ny, ph)
    # velocity maps at every snapshot: the interpolant must span tspan
    law_t = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th_true, T=-3.0)
    tV = list(ts)
    Vref = []
    for j in range(len(ts)):
        Vx, Vy, V = O.V_from_H(ref[j], B, 50.0, 50.0, ph, law_t)
        Vref.append((V, Vx, Vy))
    log_eps = 0.1 if component == "log" else None  # LossV(loss = LogSum(), component = :abs), the reference's runtests.jl:165-167
    component = "abs" if component == "log" else component
    vspec = O.LossVSpec(component=component, scale_loss=scale, log_eps=log_eps)
    law0 = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th0, T=-3.0)
    Lo, go, lam0, st = O.loss_and_grad_continuous(gl, law0, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=12),
                                                  V_ref=Vref, tV_ref=tV, vspec=vspec, loss_kind=kind, scaling=2.5)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-3.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_A_SCALAR, gpu.MLPSpec(mlp.widths, mlp.acts, None, O.POST_AFFINE, ph.minA, ph.maxA), th0)
    b.set_reference(0, ts, ref, 3)
    b.set_velocity_reference(0, tV, [v[0] for v in Vref], [v[1] for v in Vref], [v[2] for v in Vref])
    b.set_loss({"V": gpu._lib.LOSS_V, "HV": gpu._lib.LOSS_HV}[kind], component, scale, 2.5)
    b.set_velocity_loss_function(log_eps)
    Lg, gg = b.loss_grad_continuous(ts, theta=th0, reltol=1e-10, n_quadrature=12)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo), (Lg, Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (ratio, angle, relerr)
    assert rel_l2(b.lambda0(0), lam0) < 1e-5
    b.close()


@pytest.mark.parametrize("onepass", ["1", "0"])
def test_continuous_adjoint_velocity_loss_with_the_gridded_law(gpu, monkeypatch, onepass):
    """LossHV with A = NN(T) on the dual grid: the theta-part of the velocity term at the quadrature nodes goes through the
    dual-grid accumulator -- one pass (unscaled node weights, scaled into d_Gacc by k_gacc_axpy once k_vq_finish knows the
    glacier's normalisation) or the interpolate / scale / pull-back sequence (ODINN_VQ_ONEPASS=0).  Against the oracle."""
    from test_gpu_velocity import _velocity_case

    sched_env(monkeypatch, VQ_ONEPASS=onepass)
    ph = O.Phys()
    nx, ny = 64, 48
    H0, B, ts, mlp, th_true, th0, gl, cfg, ref, tV, Vref = _velocity_case(nx, ny, ph)
    S = B + H0
    T = np.asfortranarray(-3.0 - 6.5e-3 * (O.avg(S) - S.mean()))
    law_t = O.Law(kind=O.LAW_NN_A_GRIDDED, mlp=mlp, theta=th_true, T=T)
    Vref = []
    for j in range(len(ts)):
        Vx, Vy, V = O.V_from_H(ref[j], B, 50.0, 50.0, ph, law_t)
        Vref.append((V, Vx, Vy))
    vspec = O.LossVSpec(component="xy", scale_loss=True)
    law0 = O.Law(kind=O.LAW_NN_A_GRIDDED, mlp=mlp, theta=th0, T=T)
    Lo, go, lam0, st = O.loss_and_grad_continuous(gl, law0, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=12),
                                                  V_ref=Vref, tV_ref=list(ts), vspec=vspec, loss_kind="HV", scaling=2.5)
    b = gpu.GlacierBatch([(nx, ny)], [50.0])
    b.set_fields(0, H0, B)
    b.set_T_field(0, T)
    b.set_law(gpu.LAW_NN_A_GRIDDED, gpu.MLPSpec(mlp.widths, mlp.acts, None, O.POST_AFFINE, ph.minA, ph.maxA), th0)
    b.set_reference(0, ts, ref, 3)
    b.set_velocity_reference(0, ts, [v[0] for v in Vref], [v[1] for v in Vref], [v[2] for v in Vref])
    b.set_loss(gpu._lib.LOSS_HV, "xy", True, 2.5)
    Lg, gg = b.loss_grad_continuous(ts, theta=th0, reltol=1e-10, n_quadrature=12)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo), (Lg, Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (ratio, angle, relerr)
    assert rel_l2(b.lambda0(0), lam0) < 1e-5
    b.close()


def test_continuous_adjoint_needs_velocity_data_spanning_tspan(gpu):
    from test_gpu_velocity import _velocity_case

    ph = O.Phys()
    H0, B, ts, mlp, th_true, th0, gl, cfg, ref, tV, Vref = _velocity_case(64, 48, ph)
    b = gpu.GlacierBatch([(64, 48)], [50.0], T=[-3.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_A_SCALAR, gpu.MLPSpec(mlp.widths, mlp.acts, None, O.POST_AFFINE, ph.minA, ph.maxA), th0)
    b.set_reference(0, ts, ref, 3)
    b.set_velocity_reference(0, tV, [v[0] for v in Vref], [v[1] for v in Vref], [v[2] for v in Vref])  # tV = ts[2::2]
    b.set_loss(gpu._lib.LOSS_V, "xy", True, 1.0)
    with pytest.raises(Exception) as e:
        b.loss_grad_continuous(ts, theta=th0, n_quadrature=8)
    assert "span tspan" in str(e.value)
    b.close()


# ---- ContinuousVJP stencil (adjoint.jl:442-553) ------------------------------------------------
@pytest.mark.parametrize("shape,C,n", [((96, 80), 0.0, 3.0), ((65, 17), 0.0, 3.0), ((37, 40), 7e-8, 3.0),
                                       ((130, 50), 0.0, 2.6), ((3, 3), 0.0, 3.0)])
def test_continuous_vjp_kernel_matches_oracle(gpu, shape, C, n):
    nx, ny = shape
    ph = O.Phys(C=C, p=3.0, q=1.0, n=n)
    H0, B = O.synthetic_icecap(nx, ny, 100.0)
    H0 = H0 * 0.4
    rng = np.random.default_rng(7)
    lam = rng.standard_normal((nx, ny))
    b = gpu.GlacierBatch([(nx, ny)], [100.0], phys=[gpu.PhysicalParameters(**ph.__dict__)], A=[2e-17])
    b.set_fields(0, H0, B)
    b.set_vjp_method(gpu._lib.VJP_CONTINUOUS)
    law = O.Law(kind=O.LAW_CONST_A, A=2e-17)
    ref = O.vjp_H_continuous(lam, H0, B, 100.0, 100.0, ph, law)
    got = b.vjp_H(0, lam, H0)
    assert rel_l2(got, ref) < (1e-11 if n == 3.0 else 1e-10) or not ref.any()
    Hn = H0 - 30.0 * rng.uniform(size=H0.shape)  # negative thickness is clamped first (adjoint.jl:466)
    assert rel_l2(b.vjp_H(0, lam, Hn), O.vjp_H_continuous(lam, Hn, B, 100.0, 100.0, ph, law)) < 1e-10 or nx == 3
    b.set_vjp_method(gpu._lib.VJP_DISCRETE)
    assert rel_l2(b.vjp_H(0, lam, H0), O.vjp_H(lam, H0, B, 100.0, 100.0, ph, law)) < 1e-10 or nx == 3
    b.close()


@pytest.mark.parametrize("adjoint", ["discrete", "continuous"])
def test_gradients_with_continuous_vjp_match_oracle(gpu, adjoint):
    """DiscreteAdjoint(VJP_method = ContinuousVJP()) and ContinuousAdjoint(VJP_method = ContinuousVJP())
    (runtests.jl:125,141) against the oracle's loops with the same stencil."""
    nx, ny = 64, 48
    ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref = _inversion_case(gpu, nx, ny, False, k=7, step=1.0 / 48.0)
    law0 = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th0, T=-2.0)
    b = _batch(gpu, nx, ny, H0, B, gm, th0, ts, ref)
    b.set_vjp_method(gpu._lib.VJP_CONTINUOUS)
    if adjoint == "discrete":
        Lo, go, lam0 = O.loss_and_grad(gl, law0, cfg, ref, ts, vjp="continuous")
        Lg, gg = b.loss_grad(ts, theta=th0, reltol=1e-8)
    else:
        Lo, go, lam0, _ = O.loss_and_grad_continuous(gl, law0, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=16),
                                                     vjp="continuous")
        Lg, gg = b.loss_grad_continuous(ts, theta=th0, reltol=1e-8, n_quadrature=16)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (ratio, angle, relerr)
    assert rel_l2(b.lambda0(0), lam0) < 1e-5
    b.close()


@pytest.mark.parametrize("kind,arch", [("Y", "light"), ("Y", "default"), ("Y", "wide"), ("U", "default"), ("U", "light")])
def test_continuous_vjp_with_the_per_node_mlp_laws(gpu, kind, arch):
    """ContinuousVJP (adjoint.jl:442-553) with the Y law (target :D_hybrid) and the U law (target :D) -- the reference's
    'continuous adjoint with continuous VJP' for both targets (runtests.jl:178-180, 189-191): the stencil kernel against the
    oracle (alpha, beta of these laws are the reference's finite differences, hence 1e-6), and the gradient of the
    continuous adjoint with that stencil."""
    from test_gpu_parity import _mlp_pair

    ph = O.Phys()
    nx, ny = 56, 40
    H0, B = O.synthetic_alpine(nx, ny, hmax=150.0, slope=0.1)
    widths, acts = {"light": ([2, 3, 1], [1, 2]), "default": ([2, 3, 10, 3, 1], [1, 1, 1, 2]),
                    "wide": ([2, 5, 8, 20, 30, 10, 1], [3, 3, 1, 1, 1, 2])}[arch]
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-5.0])
    b.set_fields(0, H0, B)
    if kind == "Y":
        om, gm, th = _mlp_pair(gpu, widths, acts, [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
        b.set_law(gpu.LAW_NN_Y, gm, th)
        law = O.Law(kind=O.LAW_NN_Y, mlp=om, theta=th, T=-5.0)
    else:
        om, gm, th = _mlp_pair(gpu, widths, acts, [(0.0, 300.0), (0.0, 0.5)], O.POST_EXPMAX, 0.0, 50.0)
        b.set_law(gpu.LAW_NN_U, gm, th)
        law = O.Law(kind=O.LAW_NN_U, mlp=om, theta=th)
    b.set_vjp_method(gpu._lib.VJP_CONTINUOUS)
    lam = np.random.default_rng(7).standard_normal((nx, ny))
    assert rel_l2(b.vjp_H(0, lam, H0), O.vjp_H_continuous(lam, H0, B, 50.0, 50.0, ph, law)) < 1e-6
    ts = [2010.0 + j / 48.0 for j in range(4)]
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    cfg = O.SimConfig(tstops=ts, reltol=1e-8)
    ref, _, _ = O.forward(gl, law, cfg)
    ref = [r * (1.0 + 0.02 * j) for j, r in enumerate(ref)]
    b.set_reference(0, ts, ref, 3)
    Lo, go, lam0, _ = O.loss_and_grad_continuous(gl, law, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=8), vjp="continuous")
    Lg, gg = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 2e-4 and relerr < 2e-4, (ratio, angle, relerr)
    assert rel_l2(b.lambda0(0), lam0) < 2e-4
    b.close()


@pytest.mark.parametrize("kind,arch", [("Y", "light"), ("Y", "default"), ("Y", "wide"), ("A_gridded", None), ("constA_field", None)])
def test_continuous_adjoint_other_law_modes(gpu, kind, arch):
    """The reverse-ODE stage kernels of every law mode (compile-time and run-time MLP architectures
    inlined per node, hoisted gridded A, prescribed A field) against the oracle: loss, dL/dtheta,
    lambda(t0).  Both gradients (discrete and continuous adjoint) in one go."""
    from test_gpu_parity import _mlp_pair

    nx, ny = 56, 40
    ph = O.Phys()
    H0, B = O.synthetic_alpine(nx, ny, hmax=150.0, slope=0.1)
    ts = [2010.0 + j / 48.0 for j in range(4)]
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    rng = np.random.default_rng(9)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-5.0], A=[3e-17])
    b.set_fields(0, H0, B)
    if kind == "Y":
        widths, acts = {"light": ([2, 3, 1], [1, 2]), "default": ([2, 3, 10, 3, 1], [1, 1, 1, 2]),
                        "wide": ([2, 5, 8, 20, 30, 10, 1], [3, 3, 1, 1, 1, 2])}[arch]
        om, gm, th = _mlp_pair(gpu, widths, acts, [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
        b.set_law(gpu.LAW_NN_Y, gm, th)
        law = O.Law(kind=O.LAW_NN_Y, mlp=om, theta=th, T=-5.0)
        tol = 2e-4  # the Y law's alpha is a forward finite difference (target_D_hybrid.jl:58-71)
    elif kind == "A_gridded":
        om, gm, th = _mlp_pair(gpu, [1, 3, 10, 3, 1], [1, 1, 1, 2], None, O.POST_AFFINE, ph.minA, ph.maxA)
        T = np.asfortranarray(-5.0 - 4.0 * rng.uniform(size=(nx - 1, ny - 1)))
        b.set_T_field(0, T)
        b.set_law(gpu.LAW_NN_A_GRIDDED, gm, th)
        law = O.Law(kind=O.LAW_NN_A_GRIDDED, mlp=om, theta=th, T=T)
        tol = 1e-5
    else:
        Af = np.asfortranarray(3e-17 * (1.0 + 0.5 * rng.uniform(size=(nx - 1, ny - 1))))
        b.set_A_field(0, Af)
        law = O.Law(kind=O.LAW_CONST_A, A=Af)
        th = None
        tol = 1e-5
    cfg = O.SimConfig(tstops=ts, reltol=1e-8)
    ref, _, _ = O.forward(gl, law, cfg)
    ref = [r * (1.0 + 0.02 * j) for j, r in enumerate(ref)]  # something to fit
    b.set_reference(0, ts, ref, 3)
    Lo, go, lam0, _ = O.loss_and_grad_continuous(gl, law, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=8))
    Lg, gg = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo)
    assert rel_l2(b.lambda0(0), lam0) < tol
    if kind == "constA_field":
        # dL/dA on the dual grid (GriddedInv plumbing): the quadrature feeds the same accumulator;
        # its sum is the derivative w.r.t. a uniform shift of A, which is what the oracle returns for a constant-A law
        Gf = b.grad_field(0)
        assert Gf.shape == Af.shape and np.isfinite(Gf).all()
        assert abs(Gf.sum() - go[0]) <= 1e-5 * abs(go[0])
    else:
        ratio, angle, relerr = stats_err_arrays(gg, go)
        assert abs(ratio) < tol and relerr < tol, (ratio, angle, relerr)
    b.close()


def _reverse_case(gpu, case):
    """One continuous-adjoint gradient of a named case: (loss, gradient, [lambda(t0)], [(naccept, nreject)])."""
    if case == "ragged_batch":
        shapes = [(70, 57), (131, 64), (54, 46), (201, 103)]
        rng = np.random.default_rng(11)
        b = gpu.GlacierBatch(shapes, [50.0] * 4, A=[3e-17, 5e-17, 2e-17, 4e-17])
        ts = [2010.0 + j / 12.0 for j in range(5)]
        for k, (nx, ny) in enumerate(shapes):
            H0, B = O.synthetic_valley(nx, ny, 50.0)
            b.set_fields(k, H0, B)
            b.set_reference(k, ts, [H0 * (1.0 - 0.03 * j) + 0.5 * rng.random((nx, ny)) for j in range(len(ts))], 3)
        Lg, gg = b.loss_grad_continuous(ts, reltol=1e-8, n_quadrature=16)
        lam = [b.lambda0(k) for k in range(4)]
    elif case == "velocity_hv_batch":
        # LossHV on two glaciers that run out of step (per-glacier ping-pong buffers of the fused step): the velocity
        # term joins lambda at the snapshots through k_surfV_vjp<1>, which follows each glacier's current buffer
        from test_gpu_velocity import _velocity_case
        ph = O.Phys()
        shapes = [(64, 48), (80, 56)]
        b = gpu.GlacierBatch(shapes, [50.0] * 2, A=[3e-17, 6e-17])
        for k, (nx, ny) in enumerate(shapes):
            H0, B, ts, mlp, th_true, th0, gl, cfg, ref, tV, Vref = _velocity_case(nx, ny, ph)
            law_t = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th_true, T=-3.0)
            Vs = [O.V_from_H(ref[j], B, 50.0, 50.0, ph, law_t) for j in range(len(ts))]
            b.set_fields(k, H0, B)
            b.set_reference(k, ts, ref, 3)
            b.set_velocity_reference(k, ts, [v[2] for v in Vs], [v[0] for v in Vs], [v[1] for v in Vs])
        b.set_loss(gpu._lib.LOSS_HV, "xy", True, 2.5)
        Lg, gg = b.loss_grad_continuous(ts, reltol=1e-8, n_quadrature=12)
        lam = [b.lambda0(k) for k in range(2)]
    else:
        nx, ny = 96, 80
        use_mb = case == "scalar_nn_mb"
        ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref = _inversion_case(gpu, nx, ny, use_mb)
        if case == "gridded_nn":
            b = gpu.GlacierBatch([(nx, ny)], [50.0])
            b.set_fields(0, H0, B)
            S = B + H0
            Sd = 0.25 * (S[:-1, :-1] + S[1:, :-1] + S[:-1, 1:] + S[1:, 1:])
            b.set_T_field(0, np.asfortranarray(-5.0 - 6.5e-3 * (Sd - S.mean())))
            b.set_law(gpu.LAW_NN_A_GRIDDED, gm, th0)
            b.set_reference(0, ts, ref, 3)
        else:
            b = _batch(gpu, nx, ny, H0, B, gm, th0, ts, ref, mb)
        Lg, gg = b.loss_grad_continuous(ts, theta=th0, mb_times=ts[1:] if use_mb else (), reltol=1e-8, n_quadrature=16)
        lam = [b.lambda0(0)]
    res = (Lg, np.array(gg, dtype=float).ravel(), lam, [(s.naccept, s.nreject) for s in b.last_stats_rev])
    b.close()
    return res


@pytest.mark.parametrize("case", ["scalar_nn_mb", "gridded_nn", "ragged_batch", "velocity_hv_batch"])
def test_fused_reverse_step_matches_the_staged_reverse_solve(gpu, monkeypatch, case):
    """k_adj_fused_strip (the five stages of a reverse step in one kernel, face form of the H-VJP; what large
    integer-power-law batches run, forced here with ODINN_ADJ_FUSED=1) against the five k_adj_stage launches
    (ODINN_ADJ_FUSED=0): same loss, same reverse step counts, gradient and lambda(t0) equal to the tolerance of
    the adaptive reverse solve (1e-8: the two stencil forms round differently, and the embedded error estimate turns
    an ulp into a 1e-10 relative change of the step sizes; observed differences 1e-16 ... 2e-8)."""
    out = {}
    for mode in ("0", "1"):
        sched_env(monkeypatch, ADJ_FUSED=mode)
        out[mode] = _reverse_case(gpu, case)
    a, f = out["0"], out["1"]
    assert a[0] == f[0]  # the forward solve is the same code
    # accept/reject decisions sit on a threshold: an ulp of difference between the two stencil forms can flip one and
    # shift the step sequence by a few steps (seen: (76, 10) vs (72, 9) of ~80); the results still agree to the
    # tolerance of the reverse solve, asserted below
    for (na, ra), (nf, rf) in zip(a[3], f[3]):
        assert abs(na - nf) <= max(2, na // 10) and abs(ra - rf) <= max(2, ra // 2), (a[3], f[3])
    # two adaptive reverse solves at reltol = abstol = 1e-8 whose step sequences may differ by a few steps (above): they
    # agree to the integration error, a few 10 x reltol (observed 1e-16 ... 1.2e-7 over the kernel revisions)
    assert np.linalg.norm(a[1] - f[1]) <= 5e-7 * np.linalg.norm(a[1]), case
    for la, lf in zip(a[2], f[2]):
        assert rel_l2(lf, la) < 2e-6, case  # (5.4e-7 seen under ODINN_FUSED_TILES=l: another forward kernel, another step sequence)


@pytest.mark.parametrize("case", ["scalar_nn_mb", "gridded_nn", "ragged_batch", "velocity_hv_batch"])
def test_fused_reverse_step_rows_per_thread(gpu, monkeypatch, case):
    """The two instantiations of k_adj_fused_strip -- 7 rows per thread (54 x 46 output tiles) and 4 rows per thread
    (54 x 22 tiles; what batches too small to fill the GPU run, ODINN_ADJ_ROWS forces either) -- evaluate the same face
    form cell by cell; only the tiling of the error norm's partial sums differs, so the step sequences agree up to
    threshold flips and the results to the tolerance of the reverse solve."""
    sched_env(monkeypatch, ADJ_FUSED="1")
    out = {}
    # (8 rows: the register-cached instantiation of the gridded law on the forward kernel's 54 x 54 tiles; ignored otherwise)
    for rows in ("7", "4", "2") + (("8",) if case == "gridded_nn" else ()):
        sched_env(monkeypatch, ADJ_ROWS=rows)
        out[rows] = _reverse_case(gpu, case)
    for other in [r for r in out if r != "7"]:
        a, f = out["7"], out[other]
        assert a[0] == f[0]
        for (na, ra), (nf, rf) in zip(a[3], f[3]):
            assert abs(na - nf) <= max(2, na // 10) and abs(ra - rf) <= max(2, ra // 2), (a[3], f[3])
        assert np.linalg.norm(a[1] - f[1]) <= 5e-7 * np.linalg.norm(a[1]), (case, other)
        for la, lf in zip(a[2], f[2]):
            assert rel_l2(lf, la) < 2e-6, (case, other)


def test_fused_reverse_step_ice_free_shortcut_is_bitwise_exact(gpu, monkeypatch):
    """Workgroups of k_adj_fused_strip whose halo region has no ice in either bracketing snapshot skip the five
    stencil stages and run the 3S*+ update with a zero right-hand side: bit-identical to ODINN_ADJ_SKIP=0."""
    n = 320
    H0, B = O.synthetic_icecap(n, n, 100.0)
    H0 = np.asfortranarray(np.where(H0 > 500.0, H0 - 500.0, 0.0))  # small cap: most strip tiles ice-free
    ts = [0.0, 0.25, 0.5]
    sched_env(monkeypatch, ADJ_FUSED="1")
    out = []
    for skip in ("0", "1"):
        sched_env(monkeypatch, ADJ_SKIP=skip)
        b = gpu.GlacierBatch([(n, n)], [100.0], A=[4e-17])
        b.set_fields(0, H0, B)
        b.set_reference(0, ts, [H0 * (1.0 - 0.05 * j) for j in range(3)], 3)
        L, g = b.loss_grad_continuous(ts, reltol=1e-8, n_quadrature=12)
        out.append((L, np.array(g, dtype=float).ravel(), b.lambda0(0), b.last_stats_rev[0].naccept, b.last_stats_rev[0].nreject))
        b.close()
    assert out[0][0] == out[1][0] and out[0][3:] == out[1][3:]
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert np.abs(out[0][2]).max() > 0.0


def test_fused_reverse_step_with_rejected_steps(gpu, monkeypatch):
    """Loose reverse tolerances and a large dtmax: the loss jumps at the snapshot times make the reverse controller
    reject steps.  The fused kernel then simply repeats from the untouched lambda[cur] (no saved copy); the staged
    path restarts from its uprev register.  Same accept / reject counts, gradients equal to rounding."""
    out = {}
    for mode in ("0", "1"):
        sched_env(monkeypatch, ADJ_FUSED=mode)
        shapes = [(130, 97), (96, 80)]
        b = gpu.GlacierBatch(shapes, [50.0] * 2, A=[6e-17, 4e-17])
        ts = [2010.0, 2010.5, 2011.0, 2011.5]
        for k, (nx, ny) in enumerate(shapes):
            H0, B = O.synthetic_valley(nx, ny, 50.0)
            b.set_fields(k, H0, B)
            b.set_reference(k, ts, [H0 * (1.0 - 0.1 * j) for j in range(len(ts))], 3)
        L, g = b.loss_grad_continuous(ts, reltol=1e-8, adj_reltol=1e-4, adj_abstol=1e-6, adj_dtmax=0.5, n_quadrature=6)
        out[mode] = (L, np.array(g, dtype=float).ravel(), [(s.naccept, s.nreject) for s in b.last_stats_rev], b.lambda0(0))
        b.close()
    a, f = out["0"], out["1"]
    assert a[2] == f[2] and sum(r for _, r in a[2]) > 0, (a[2], f[2])
    assert np.abs(a[1] - f[1]).max() <= 1e-10 * np.abs(a[1]).max()
    # (lambda(t0) of a reverse solve run at reltol 1e-4 across three jumps: 1e-11 under the default schedule, 9e-10 with the LDS-tile
    #  forward kernel -- ODINN_FUSED_TILES=l in tools/suite_matrix.sh -- whose snapshots differ from the strip kernel's in the last bits)
    assert rel_l2(f[3], a[3]) < 1e-8


@pytest.mark.parametrize("case", ["scalar_nn_mb", "gridded_nn", "ragged_batch", "rejections", "mb_only_stops", "y_table"])
@pytest.mark.parametrize("rows", ["7", "4", "2"])
def test_self_controlled_reverse_step_matches_the_three_launch_loop(gpu, monkeypatch, case, rows):
    """ODINN_ADJ_SC=1: every workgroup of the fused reverse step decides the previous attempt of its glacier itself (error norm,
    PID controller, stop tables, the AdjState of the coming step) and a launch that follows a step onto a snapshot time is that
    glacier's post-step (loss cotangent, mass-balance VJP) -- no k_controller / k_adj_poststep launches.  Same kernel arithmetic
    and the same decisions as the three-launch loop (ODINN_ADJ_SC=0): same reverse step counts, bit-identical loss, gradient
    and lambda(t0) -- on a single glacier with a mass balance, the gridded law (dual-grid accumulator fed by stage 1), a ragged
    batch whose glaciers run out of step, a reverse solve with rejected steps, and mass-balance times that are not result
    stops."""
    sched_env(monkeypatch, ADJ_FUSED="1")
    sched_env(monkeypatch, ADJ_ROWS=rows)
    out = {}
    for sc in ("0", "1"):
        sched_env(monkeypatch, ADJ_SC=sc)
        if case == "rejections":
            shapes = [(130, 97), (96, 80)]
            b = gpu.GlacierBatch(shapes, [50.0] * 2, A=[6e-17, 4e-17])
            ts = [2010.0, 2010.5, 2011.0, 2011.5]
            for k, (nx, ny) in enumerate(shapes):
                H0, B = O.synthetic_valley(nx, ny, 50.0)
                b.set_fields(k, H0, B)
                b.set_reference(k, ts, [H0 * (1.0 - 0.1 * j) for j in range(len(ts))], 3)
            L, g = b.loss_grad_continuous(ts, reltol=1e-8, adj_reltol=1e-4, adj_abstol=1e-6, adj_dtmax=0.5, n_quadrature=6)
            out[sc] = (L, np.array(g, dtype=float).ravel(), [b.lambda0(k) for k in range(2)],
                       [(s.naccept, s.nreject) for s in b.last_stats_rev])
            b.close()
        elif case == "y_table":
            # the Y law through its table, `:Linear` gradient: stage 1 of the fused step emits the node pairs of a quadrature node
            # (both loops), the sort-free contraction runs on the lanes
            from test_gpu_parity import _mlp_pair
            sched_env(monkeypatch, LAW_TABLE=None)  # (the case IS the table: tools/suite_matrix.sh runs the suite with it off)
            ph = O.Phys()
            om, gm, th = _mlp_pair(gpu, [2, 3, 10, 3, 1], [1, 1, 1, 2], [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
            shapes = [(70, 57), (131, 64), (54, 46)]
            b = gpu.GlacierBatch(shapes, [50.0] * 3, T=[-5.0, -11.0, -2.0])
            ts = [2010.0 + j / 24.0 for j in range(4)]
            for k, (nx, ny) in enumerate(shapes):
                H0, B = O.synthetic_alpine(nx, ny, hmax=150.0)
                b.set_fields(k, H0, B)
                b.set_reference(k, ts, [H0 * (1.0 - 0.01 * j) for j in range(len(ts))], 3)
            b.set_law(gpu.LAW_NN_Y, gm, th)
            assert b.law_table()["usable"]
            L, g = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=12)
            out[sc] = (L, np.array(g, dtype=float).ravel(), [b.lambda0(k) for k in range(3)],
                       [(s.naccept, s.nreject) for s in b.last_stats_rev])
            b.close()
        elif case == "mb_only_stops":
            nx, ny = 96, 80
            ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref = _inversion_case(gpu, nx, ny, True)
            b = _batch(gpu, nx, ny, H0, B, gm, th0, ts, ref, mb)
            # mass-balance times between the result stops (and on them)
            mbt = sorted(set(list(ts[1:]) + [0.5 * (ts[j] + ts[j + 1]) for j in range(len(ts) - 1)]))
            L, g = b.loss_grad_continuous(ts, theta=th0, mb_times=mbt, reltol=1e-8, n_quadrature=16)
            out[sc] = (L, np.array(g, dtype=float).ravel(), [b.lambda0(0)], [(s.naccept, s.nreject) for s in b.last_stats_rev])
            b.close()
        else:
            out[sc] = _reverse_case(gpu, case)
    a, f = out["0"], out["1"]
    if case == "rejections":
        assert sum(r for _, r in a[3]) > 0
    assert a[3] == f[3], (a[3], f[3])
    assert a[0] == f[0]
    assert np.array_equal(a[1], f[1]), np.abs(a[1] - f[1]).max() / np.abs(a[1]).max()
    for la, lf in zip(a[2], f[2]):
        assert np.array_equal(la, lf)
    assert np.abs(a[1]).max() > 0.0
