"""Pins the oracle's surface-velocity restatement with the reference's own test
(test_adjoint_surface_V, test/SIA2D_adjoint.jl:209-330; thresholds [2e-4, 2e-4, 2e-2])
and LossV / LossHV gradients against finite differences."""
import numpy as np
import pytest

from conftest import stats_err_arrays
from oracle import sia2d_oracle as O


def test_surface_V_vjps_vs_fd():
    # C = 0 as in the reference's test; with sliding the reference's dVelocity^/dH and /dgradH
    # (target_A.jl:110-142) are not the derivatives of its Velocity^ (:94-108) -- they are restated
    # as written (GPU == oracle is tested with C > 0), so no FD claim is made there.
    ph = O.Phys(maxA=8e-18)
    H0, B = O.synthetic_alpine(36, 31)
    rng = np.random.default_rng(1234)
    mlp = O.default_nn(1, light=True, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    th = mlp.init_theta(rng)
    law = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th, T=-5.0)
    w1, w2 = rng.standard_normal(H0.shape), rng.standard_normal(H0.shape)

    def f(H, t=th):
        vx, vy = O.surface_V(H, B, 50.0, 50.0, ph, law, t)
        return np.sum(vx * O.inn1(w1) + vy * O.inn1(w2))

    g = O.vjp_surface_V_H(w1, w2, H0, B, 50.0, 50.0, ph, law)
    best = [np.inf] * 3
    f0 = f(H0)
    for eps in (1e-3, 1e-5, 1e-7):
        gn = np.zeros_like(H0)
        for i in range(H0.shape[0]):
            for j in range(H0.shape[1]):
                Hp = H0.copy()
                Hp[i, j] += eps
                gn[i, j] = (f(Hp) - f0) / eps
        best = [min(a, abs(s)) for a, s in zip(best, stats_err_arrays(g, gn))]
    assert best[0] < 2e-4 and best[1] < 2e-4 and best[2] < 2e-2, best
    gt = O.vjp_surface_V_theta(w1, w2, H0, B, 50.0, 50.0, ph, law)
    gn = np.zeros_like(th)
    for q in range(th.size):
        e = np.zeros_like(th)
        e[q] = 1e-6
        gn[q] = (f(H0, th + e) - f(H0, th - e)) / 2e-6
    ratio, angle, relerr = stats_err_arrays(gt, gn)
    assert abs(ratio) < 2e-4 and abs(angle) < 2e-4 and relerr < 2e-2


def test_V_from_H_inn1_pairing():
    ph = O.Phys()
    H0, B = O.synthetic_alpine(20, 17)
    law = O.Law(kind=O.LAW_CONST_A, A=2e-17)
    Vx, Vy, V = O.V_from_H(H0, B, 50.0, 50.0, ph, law)
    vx, vy = O.surface_V(H0, B, 50.0, 50.0, ph, law)
    assert Vx.shape == H0.shape and np.array_equal(Vx[:-1, :-1], vx) and np.all(Vx[-1, :] == 0) and np.all(Vy[:, -1] == 0)
    assert np.allclose(V, np.hypot(Vx, Vy))


@pytest.mark.parametrize("kind,component,scale", [("V", "xy", True), ("V", "abs", True), ("HV", "xy", False)])
def test_velocity_losses_gradient_vs_fd(kind, component, scale):
    ph = O.Phys()
    H0, B = O.synthetic_alpine(40, 33, hmax=160.0, slope=0.1)
    ts = [2010.0 + j / 96.0 for j in range(7)]
    mlp = O.default_nn(1, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    th_true, th0 = mlp.init_theta(np.random.default_rng(42)), mlp.init_theta(np.random.default_rng(1234))
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    cfg = O.SimConfig(tstops=ts, reltol=1e-10)
    law_t = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th_true, T=-3.0)
    ref, _, _ = O.forward(gl, law_t, cfg)
    tV = ts[2::2]
    Vref = []
    for t in tV:
        Vx, Vy, V = O.V_from_H(ref[ts.index(t)], B, 50.0, 50.0, ph, law_t)
        Vref.append((V, Vx, Vy))
    vs = O.LossVSpec(component, scale)
    L, g, _ = O.loss_and_grad_HV(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th0, T=-3.0), cfg, ref, ts, Vref, tV, vs,
                                 loss_kind=kind, scaling=2.5)

    def loss_at(th):
        return O.loss_and_grad_HV(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th, T=-3.0), cfg, ref, ts, Vref, tV, vs,
                                  loss_kind=kind, scaling=2.5)[0]

    idx = np.arange(0, g.size, 11)
    gn = np.zeros_like(g)
    for q in idx:
        e = np.zeros_like(g)
        e[q] = 1e-4
        gn[q] = (loss_at(th0 + e) - loss_at(th0 - e)) / 2e-4
    ratio, angle, relerr = stats_err_arrays(g[idx], gn[idx])
    assert abs(ratio) < 1e-2 and abs(angle) < 1e-7 and relerr < 1e-2, (ratio, angle, relerr)


def test_continuous_adjoint_with_lossV_vs_fd():
    """ContinuousAdjoint + LossV (runtests.jl:163-166, thresholds [1e-2, 1e-5, 1e-2]): the theta-part of the
    velocity loss is a time INTEGRAL (quadrature, Delta-t = 1) while the loss is the discrete Delta-t-weighted
    sum, so the two agree to O(Delta-t): 1.2 % at this snapshot spacing -- own bound 2e-2, stated."""
    ph = O.Phys()
    H0, B = O.synthetic_alpine(40, 33, hmax=160.0, slope=0.1)
    ts = [2010.0 + j / 96.0 for j in range(7)]
    mlp = O.default_nn(1, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    th_true, th0 = mlp.init_theta(np.random.default_rng(42)), mlp.init_theta(np.random.default_rng(1234))
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    cfg = O.SimConfig(tstops=ts, reltol=1e-10)
    law_t = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th_true, T=-3.0)
    ref, _, _ = O.forward(gl, law_t, cfg)
    Vref = []
    for j in range(len(ts)):
        Vx, Vy, V = O.V_from_H(ref[j], B, 50.0, 50.0, ph, law_t)
        Vref.append((V, Vx, Vy))
    vs = O.LossVSpec("xy", True)
    law0 = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th0, T=-3.0)
    L, g, _, _ = O.loss_and_grad_continuous(gl, law0, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=24),
                                            V_ref=Vref, tV_ref=ts, vspec=vs, loss_kind="V")
    Ld, gd, _ = O.loss_and_grad_HV(gl, law0, cfg, ref, ts, Vref, ts, vs, loss_kind="V")
    assert abs(L - Ld) <= 1e-12 * abs(Ld)

    def loss_at(th):
        return O.loss_and_grad_HV(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th, T=-3.0), cfg, ref, ts, Vref, ts, vs,
                                  loss_kind="V")[0]

    idx = np.arange(0, g.size, 11)
    gn = np.zeros_like(g)
    for q in idx:
        e = np.zeros_like(g)
        e[q] = 1e-4
        gn[q] = (loss_at(th0 + e) - loss_at(th0 - e)) / 2e-4
    ratio, angle, relerr = stats_err_arrays(g[idx], gn[idx])
    assert abs(ratio) < 2e-2 and abs(angle) < 1e-5 and relerr < 2e-2, (ratio, angle, relerr)
    with pytest.raises(ValueError):  # velocity data that do not span tspan
        O.loss_and_grad_continuous(gl, law0, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=4),
                                   V_ref=Vref[2::2], tV_ref=ts[2::2], vspec=vs, loss_kind="V")


def test_D_hybrid_velocity_is_restated_as_written_not_as_consistent():
    """Target :D_hybrid's Velocity^ (target_D_hybrid.jl:210-372) mixes Gamma = 2 (rho g)^n / (n + 2) in the value with
    Gamma^ = 2 (rho g)^n / (n + 1) in the theta-weight: the restated dVelocity^/dtheta is (n + 2) / (n + 1) times the true
    derivative of the restated value.  This test pins that the oracle reproduces the reference's text, not a corrected one."""
    ph = O.Phys()
    om = O.MLP([2, 3, 10, 3, 1], [1, 1, 1, 2], [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
    th = om.init_theta(np.random.default_rng(9))
    H, B = O.synthetic_valley(48, 40, 50.0)
    rng = np.random.default_rng(1)
    w1, w2 = rng.standard_normal(H.shape), rng.standard_normal(H.shape)
    law = O.Law(kind=O.LAW_NN_Y, mlp=om, theta=th, T=-5.0, interpolation="none")
    g = O.vjp_surface_V_theta(w1, w2, H, B, 50.0, 50.0, ph, law)

    def f(t):
        vx, vy, _ = O.V_from_H(H, B, 50.0, 50.0, ph, O.Law(kind=O.LAW_NN_Y, mlp=om, theta=t, T=-5.0))
        return np.sum(vx * w1 + vy * w2)

    for q in (3, 17, 40):
        e = np.zeros_like(th)
        e[q] = 1e-6
        fd = (f(th + e) - f(th - e)) / 2e-6
        assert abs(fd / g[q] - (ph.n + 1.0) / (ph.n + 2.0)) < 1e-5, (q, fd, g[q])
