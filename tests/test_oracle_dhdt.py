"""-m "not gpu": the oracle's restatement of LossDhdt, a time-aggregated loss (src/losses/TimeAggregatedLosses.jl:38-113;
its place in both adjoints: src/inverse/SIA2D/gradient.jl:170-215, :369-449), against finite differences of the loss."""
import numpy as np

from conftest import stats_err_arrays
from oracle import sia2d_oracle as O
from test_oracle_gradient import _case


def _fd(f, th0, idx, eps=1e-4):
    gn = np.zeros_like(th0)
    for q in idx:
        e = np.zeros_like(th0)
        e[q] = eps
        gn[q] = (f(th0 + e) - f(th0 - e)) / (2 * eps)
    return gn


def test_dhdt_terms_definition():
    ph, gl, mlp, th_true, th0, ts, cfg, ref = _case(1.0 / 480.0, 13)
    snaps, _, _ = O.forward(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th0, T=-2.0), cfg)
    cfg.dhdt, cfg.dhdt_weight = (ts[2], ts[12], -3.0), 2.5
    l, terms = O.dhdt_loss_terms(snaps, ts, cfg)
    H0, H1 = snaps[2], snaps[12]
    m = H0 > 1e-2
    dh = (H1[m] - H0[m]).mean() / (ts[12] - ts[2])
    assert np.isclose(l, 2.5 * (dh + 3.0) ** 2, rtol=1e-14)
    assert set(terms) == {2, 12} and np.array_equal(terms[2], -terms[12])
    assert np.array_equal(terms[12] != 0, m)
    # the fields are the derivative of the loss w.r.t. H1 (mask fixed)
    e = np.zeros_like(H1)
    e[m.nonzero()[0][5], m.nonzero()[1][5]] = 1e-3
    s2 = list(snaps)
    s2[12] = H1 + e
    l2, _ = O.dhdt_loss_terms(s2, ts, cfg)
    assert np.isclose((l2 - l) / 1e-3, (terms[12] * (e != 0)).sum(), rtol=1e-3)


def test_dhdt_gradient_vs_finite_differences_both_adjoints():
    """LossH + 4 LossDhdt, and LossDhdt alone: dL/dtheta of the discrete adjoint (reference bound for LossH alone:
    [5e-3, 1e-8, 5e-3], runtests.jl:116-117; own bound here) and of the continuous adjoint ([1e-3, 1e-8, 1e-3], :127)."""
    ph, gl, mlp, th_true, th0, ts, cfg, ref = _case(1.0 / 480.0, 13)
    law = lambda th: O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th, T=-2.0)
    snaps_true, _, _ = O.forward(gl, law(th_true), cfg)
    m = snaps_true[2] > 1e-2
    dh_true = (snaps_true[12][m] - snaps_true[2][m]).mean() / (ts[12] - ts[2])
    idx = np.arange(0, th0.size, 6)
    for with_H in (True, False):
        cfg.dhdt, cfg.dhdt_weight = (ts[2], ts[12], dh_true), (4.0 if with_H else 1.0)
        Href, tH = (ref, ts) if with_H else ([], [])

        def loss_at(th):
            s, _, _ = O.forward(gl, law(th), cfg)
            return (O.loss_H(s, ts, Href, tH, 3) if with_H else 0.0) + O.dhdt_loss_terms(s, ts, cfg)[0]

        gn = _fd(loss_at, th0, idx)
        L, g, _ = O.loss_and_grad(gl, law(th0), cfg, Href, tH)
        assert np.isclose(L, loss_at(th0), rtol=1e-12)
        ratio, angle, relerr = stats_err_arrays(g[idx], gn[idx])
        assert abs(ratio) < 2e-2 and abs(angle) < 1e-7 and relerr < 2e-2, (with_H, ratio, angle, relerr)
        Lc, gc, _, _ = O.loss_and_grad_continuous(gl, law(th0), cfg, Href, tH, O.ContinuousAdjointCfg(n_quadrature=200))
        assert np.isclose(Lc, L, rtol=1e-12)
        ratio, angle, relerr = stats_err_arrays(gc[idx], gn[idx])
        # lambda jumps at t0 inside the quadrature interval: the Gauss-Legendre sum converges slowly there (1.4e-2 with 40 nodes)
        assert abs(ratio) < 5e-3 and abs(angle) < 1e-7 and relerr < 5e-3, (with_H, ratio, angle, relerr)
    cfg.dhdt = None
