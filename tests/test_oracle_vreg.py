"""-m "not gpu": the oracle's restatement of VelocityRegularization (src/losses/Regularization.jl:64-79,192-245; the term
of the reference's documented multi-objective example MultiLoss((LossH(), VelocityRegularization()), ...), tested there
by test/runtests.jl:213-221) against its definition and finite differences of the loss in both adjoints."""
import numpy as np

from conftest import stats_err_arrays
from oracle import sia2d_oracle as O
from test_oracle_dhdt import _fd
from test_oracle_gradient import _case


def test_vreg_definition_and_state_derivative():
    ph, gl, mlp, th_true, th0, ts, cfg, ref = _case(1.0 / 480.0, 13)
    law = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th0, T=-2.0)
    snaps, _, _ = O.forward(gl, law, cfg)
    H = snaps[6]
    l, gH, gth = O.vreg_backward(H, gl, law, None, 3)
    Vx, Vy, V = O.V_from_H(H, gl.B, gl.dx, gl.dy, ph, law)
    m = O.is_in_glacier(H, 3) & (V > 0)
    assert m.sum() > 50 and np.isclose(l, (O.laplacian(V, gl.dx, gl.dy)[m] ** 2).sum(), rtol=1e-14)
    # derivative w.r.t. the state with the mask held fixed (perturbation inside the mask's support, small enough)
    rng = np.random.default_rng(1)
    e = rng.standard_normal(H.shape) * O.is_in_glacier(H, 5)
    eps = 1e-4

    def reg(Hp):
        v = O.V_from_H(Hp, gl.B, gl.dx, gl.dy, ph, law)[2]
        return (O.laplacian(v, gl.dx, gl.dy)[m] ** 2).sum()

    fd = (reg(H + eps * e) - reg(H - eps * e)) / (2 * eps)
    assert np.isclose(fd, (gH * e).sum(), rtol=1e-6)


def test_vreg_gradient_vs_finite_differences_both_adjoints():
    """'Just regularization' and 'Empirical and regularization' of runtests.jl:213-221 (thresholds there: [1e-2, 1e-8, 1e-2]
    and [1e-4, 1e-8, 1e-4] for the ContinuousAdjoint)."""
    ph, gl, mlp, th_true, th0, ts, cfg, ref = _case(1.0 / 480.0, 13)
    law = lambda th: O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th, T=-2.0)
    idx = np.arange(0, th0.size, 6)
    cfg.vreg_times, cfg.vreg_distance = list(ts), 3
    for with_H, lam_reg in ((False, 1e2), (True, 2.0)):
        cfg.vreg_weight = lam_reg
        Href, tH = (ref, ts) if with_H else ([], [])

        def loss_at(th):
            s, _, _ = O.forward(gl, law(th), cfg)
            return (O.loss_H(s, ts, Href, tH, 3) if with_H else 0.0) + O.vreg_loss_terms(s, ts, cfg, gl, law(th))[0]

        gn = _fd(loss_at, th0, idx)
        L, g, _ = O.loss_and_grad(gl, law(th0), cfg, Href, tH)
        assert np.isclose(L, loss_at(th0), rtol=1e-12)
        ratio, angle, relerr = stats_err_arrays(g[idx], gn[idx])
        assert abs(ratio) < 3e-2 and abs(angle) < 1e-5 and relerr < 3e-2, (with_H, ratio, angle, relerr)
        Lc, gc, _, _ = O.loss_and_grad_continuous(gl, law(th0), cfg, Href, tH, O.ContinuousAdjointCfg(n_quadrature=200))
        assert np.isclose(Lc, L, rtol=1e-12)
        ratio, angle, relerr = stats_err_arrays(gc[idx], gn[idx])
        # the loss is a Riemann sum over the data times (first one with weight 0), the theta-part of the ContinuousAdjoint
        # the Gauss-Legendre integral of the same integrand (gradient.jl:475-503): they differ at O(data spacing) -- the
        # direction agrees to 1e-6, the magnitude to 2 % with 12 data intervals
        assert abs(ratio) < 3e-2 and abs(angle) < 1e-5 and relerr < 3e-2, (with_H, ratio, angle, relerr)
    cfg.vreg_weight = 0.0
