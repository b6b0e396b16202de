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
