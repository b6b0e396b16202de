"""End-to-end pins of the oracle's discrete adjoint (gradient.jl:129-275): gradient vs
finite differences of the forward loss (test_grad_finite_diff, thresholds
[5e-3, 1e-8, 5e-3] runtests.jl:116-117), golden-vector regression, recovery of A
(test/inversion_test.jl:154-163)."""
import os

import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays
from oracle import sia2d_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _case(step, k, use_mb=False, nx=48, ny=40):
    ph = O.Phys()
    H0, B = O.synthetic_valley(nx, ny, 50.0)
    ts = [2010.0 + j * step for j in range(k)]
    mlp = O.default_nn(1, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    th_true = mlp.init_theta(np.random.default_rng(42))
    th0 = mlp.init_theta(np.random.default_rng(1234))
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    mb = None
    if use_mb:
        S0 = B + H0
        mb = O.MassBalance(mb0=6e-3 * (S0 - np.percentile(S0[H0 > 0], 60)) * step, dmb_dS=6e-3 * step, S_ref=S0, mb_max=1.2 * step)
    cfg = O.SimConfig(tstops=ts, reltol=1e-10, mb=mb, mb_times=ts[1:] if use_mb else ())
    ref, _, _ = O.forward(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th_true, T=-2.0), cfg)
    return ph, gl, mlp, th_true, th0, ts, cfg, ref


@pytest.mark.parametrize("use_mb", [False, True])
def test_discrete_adjoint_vs_finite_differences(use_mb):
    ph, gl, mlp, th_true, th0, ts, cfg, ref = _case(1.0 / 480.0, 13, use_mb)
    L, g, _ = O.loss_and_grad(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th0, T=-2.0), cfg, ref, ts)

    def loss_at(th):
        s, _, _ = O.forward(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th, T=-2.0), cfg)
        return O.loss_H(s, ts, ref, ts, 3)

    idx = np.arange(0, g.size, 6)
    gn = np.zeros_like(g)
    for q in idx:
        e = np.zeros_like(g)
        e[q] = 1e-4
        gn[q] = (loss_at(th0 + e) - loss_at(th0 - e)) / 2e-4
    ratio, angle, relerr = stats_err_arrays(g[idx], gn[idx])
    # reference thresholds without MB (runtests.jl:116-117); with the (non-smooth) MB mask the
    # reference has no DiscreteAdjoint threshold -- own bound, stated
    thr = (1e-2, 1e-7, 1e-2) if use_mb else (5e-3, 1e-8, 5e-3)
    assert abs(ratio) < thr[0] and abs(angle) < thr[1] and relerr < thr[2], (ratio, angle, relerr)


def test_golden_rhs_vectors_unchanged():
    from tests_golden_loader import load_make_golden

    mg = load_make_golden()
    for c in mg.CASES:
        ref = np.load(os.path.join(GOLD, f"rhs_{c}.npz"))
        now = mg.compute(c)
        for key in ("dH", "vjp_H", "vjp_theta"):
            assert rel_l2(now[key], ref[key]) < 1e-13, (c, key)


def test_golden_solve_unchanged():
    from tests_golden_loader import load_make_golden

    mg = load_make_golden()
    ref = np.load(os.path.join(GOLD, "solve_valley_nnA.npz"))
    now = mg.solve_case()
    assert rel_l2(now["ref"], ref["ref"]) < 1e-11
    assert abs(now["loss"] - ref["loss"]) <= 1e-9 * abs(ref["loss"])
    assert rel_l2(now["grad"], ref["grad"]) < 1e-8


def test_inversion_recovers_A():
    """Functional inversion on the oracle alone: L-BFGS on (loss, discrete-adjoint gradient)
    drives the loss down by >= 1e-6x and recovers A to < 1e-3 relative
    (test/inversion_test.jl:154-163)."""
    from scipy.optimize import minimize

    ph, gl, mlp, th_true, th0, ts, cfg, ref = _case(1.0 / 240.0, 7, False, nx=40, ny=32)
    A_true = O.law_value(O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th_true, T=-2.0), ph, None, None)
    hist = []

    def fg(th):
        L, g, _ = O.loss_and_grad(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th, T=-2.0), cfg, ref, ts)
        hist.append(L)
        return L, g

    res = minimize(fg, th0, jac=True, method="L-BFGS-B", options=dict(maxiter=60, ftol=0, gtol=0))
    A_fit = O.law_value(O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=res.x, T=-2.0), ph, None, None)
    assert min(hist) < 1e-6 * hist[0]
    assert abs(A_fit - A_true) / A_true < 1e-3


def test_continuous_adjoint_vs_finite_differences():
    """Oracle restatement of ContinuousAdjoint(VJP_method = DiscreteVJP()) (gradient.jl:276-539) vs
    central finite differences with the reference's thresholds [1e-3, 1e-8, 1e-3] (runtests.jl:127),
    monthly snapshots."""
    ph, gl, mlp, th_true, th0, ts, cfg, ref = _case(1.0 / 12.0, 7, False)
    law = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th0, T=-2.0)
    L, g, lam0, st = O.loss_and_grad_continuous(gl, law, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=60))

    def loss_at(th):
        s, _, _ = O.forward(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th, T=-2.0), cfg)
        return O.loss_H(s, ts, ref, ts, 3)

    idx = np.arange(0, g.size, 6)
    gn = np.zeros_like(g)
    for q in idx:
        e = np.zeros_like(g)
        e[q] = 1e-4
        gn[q] = (loss_at(th0 + e) - loss_at(th0 - e)) / 2e-4
    ratio, angle, relerr = stats_err_arrays(g[idx], gn[idx])
    assert abs(ratio) < 1e-3 and abs(angle) < 1e-8 and relerr < 1e-3, (ratio, angle, relerr)
    assert abs(L - loss_at(th0)) <= 1e-12 * L


def test_gauss_quadrature_nodes():
    """GaussQuadrature (gradient.jl:560-566): exact for polynomials up to degree 2n-1, weights sum to t1-t0."""
    x, w = O.gauss_quadrature(2010.0, 2012.0, 7)
    assert abs(w.sum() - 2.0) < 1e-14
    for d in range(14):
        exact = ((2012.0 - 2011.0) ** (d + 1) - (2010.0 - 2011.0) ** (d + 1)) / (d + 1)
        assert abs(np.sum(w * (x - 2011.0) ** d) - exact) < 5e-12  # ulp(2011) = 2.3e-13
