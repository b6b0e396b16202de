"""Pins the oracle's time integration: RDPK3Sp35 coefficient set (order conditions),
convergence order, the PID controller, the Halfar known answer and volume conservation."""
import math

import numpy as np
import pytest

from conftest import rel_l2
from oracle import sia2d_oracle as O


def _butcher():
    """Butcher tableau implied by the 3S*+ recurrence, in 40-digit arithmetic."""
    from mpmath import mp, mpf

    mp.dps = 40
    txt = open(O.__file__).read()

    def tup(name):
        s = txt[txt.index(name + " = ("):]
        s = s[s.index("(") + 1: s.index(")")]
        return [mpf(v.strip()) for v in s.replace("\n", " ").split(",") if v.strip()]

    g1, g2, g3, dl, bt, c, bh = (tup(n) for n in ("RDPK_G1", "RDPK_G2", "RDPK_G3", "RDPK_DELTA", "RDPK_BETA", "RDPK_C", "RDPK_BHAT"))
    e = lambda i: [mpf(1 if k == i else 0) for k in range(6)]
    lin = lambda *ts: [sum(a * v[k] for a, v in ts) for k in range(6)]
    up = e(0)
    tmp = up[:]
    u = lin((1, tmp), (bt[0], e(1)))
    Y = [up[:], u[:]]
    for i in range(1, 5):
        tmp = lin((1, tmp), (dl[i], u))
        u = lin((g1[i], u), (g2[i], tmp), (g3[i], up), (bt[i], e(i + 1)))
        Y.append(u[:])
    A = [[Y[i][j + 1] for j in range(5)] for i in range(5)]
    b = [Y[5][j + 1] for j in range(5)]
    return A, b, c, bh, [y[0] for y in Y]


def test_rdpk3sp35_order_conditions():
    from mpmath import mpf

    A, b, c, bh, u0 = _butcher()
    cs = [sum(r) for r in A]
    assert all(abs(x - 1) < mpf("1e-35") for x in u0)  # consistency: every stage reproduces constants
    assert all(abs(cs[i] - c[i]) < mpf("1e-33") for i in range(5))  # row sums == published c
    conds = [sum(b) - 1, sum(b[i] * cs[i] for i in range(5)) - mpf(1) / 2,
             sum(b[i] * cs[i] ** 2 for i in range(5)) - mpf(1) / 3,
             sum(b[i] * A[i][j] * cs[j] for i in range(5) for j in range(5)) - mpf(1) / 6]
    assert all(abs(x) < mpf("1e-33") for x in conds), conds  # third order
    assert abs(sum(bh) - 1) < mpf("1e-33")  # embedded weights consistent
    assert abs(sum(bh[i] * cs[i] for i in range(5)) - mpf(1) / 2) < mpf("1e-6")  # second order (to optimiser tol)


def test_convergence_order_3():
    f = lambda u: np.array([u[1], -u[0]])  # harmonic oscillator
    u0 = np.array([1.0, 0.0])
    errs = []
    for n in (20, 40, 80):
        u = u0.copy()
        for _ in range(n):
            u, _ = O.rdpk3sp35_step(f, u, 1.0 / n)
        errs.append(np.linalg.norm(u - np.array([math.cos(1.0), -math.sin(1.0)])))
    p1, p2 = math.log2(errs[0] / errs[1]), math.log2(errs[1] / errs[2])
    assert 2.8 < p1 < 3.3 and 2.8 < p2 < 3.3, (errs, p1, p2)


def test_adaptive_solve_tolerance_and_tstops():
    f = lambda u: -u * u
    sn, st, _ = O.solve(f, np.array([1.0, 2.0]), [0.0, 0.3, 1.0, 2.5], reltol=1e-8, abstol=1e-10)
    for t, s in zip([0.0, 0.3, 1.0, 2.5], sn):
        exact = np.array([1.0, 2.0]) / (1.0 + np.array([1.0, 2.0]) * t)
        assert np.allclose(s, exact, rtol=2e-7)
    sn2, st2, _ = O.solve(f, np.array([1.0, 2.0]), [0.0, 2.5], reltol=1e-4, abstol=1e-6)
    assert st2.naccept < st.naccept  # looser tolerance, fewer steps
    # callback applied exactly at its stop and included in the stored state
    sn3, _, inc = O.solve(f, np.array([1.0]), [0.0, 1.0, 2.0], reltol=1e-8, callback=lambda u, t: u + 1.0, callback_times=[1.0])
    assert abs(sn3[1][0] - (0.5 + 1.0)) < 1e-5 and abs(inc[1.0][0] - 1.0) < 1e-12


def test_step_size_collapse_ends_the_solve_with_dtmin():
    """An error tolerance that no step size can meet: every attempt is rejected (the controller's factor bottoms out at 1 - pi / 4) and
    the solve is given up after 256 attempts in a row that did not advance t, instead of spinning until maxiters -- the device's rule
    (ODINN_ERR_DTMIN, tests/test_gpu_classical_errors.py); OrdinaryDiffEq's check_error ends such a solve earlier, at the first
    dt <= dtmin = eps(t) (ReturnCode.DtLessThanMin)."""
    f = lambda u: -u * u
    with pytest.raises(RuntimeError, match="dtmin"):
        O.solve(f, np.array([1.0, 2.0]), [2010.0, 2010.5], reltol=1e-30, abstol=1e-300)
    sn, st, _ = O.solve(f, np.array([1.0, 2.0]), [2010.0, 2010.5], reltol=1e-8, abstol=1e-10)  # (the same problem at a tolerance it can meet)
    assert st.naccept > 0 and np.allclose(sn[-1], np.array([1.0, 2.0]) / (1.0 + np.array([1.0, 2.0]) * 0.5), rtol=2e-7)


def test_halfar_and_volume():
    """60x60 dome, R0=2000, H0=400, A=1.1e-17, 10 years.  Expect dome
    height within ~0.1 m of the similarity solution, rel-L2 ~1e-2 (margin resolution) and
    volume conserved to rounding."""
    nx = ny = 60
    R0, H00, A = 2000.0, 400.0, 1.1e-17
    dx = R0 / nx / 0.4
    x = (np.arange(nx) - nx / 2 + 0.5) * dx
    X, Y = np.meshgrid(x, x, indexing="ij")
    t0 = O.halfar_t0(A, H00, R0)
    H0 = O.halfar(X, Y, t0, A, H00, R0)
    assert abs(H0.max() - H00) < 2.0  # no cell centre sits exactly on the summit
    ph = O.Phys()
    law = O.Law(kind=O.LAW_CONST_A, A=A)
    B = np.zeros_like(H0)
    f = lambda H: O.sia2d_rhs(H, B, dx, dx, ph, law)
    sn, st, _ = O.solve(f, H0, [t0, t0 + 10.0], reltol=1e-8)
    exact = O.halfar(X, Y, t0 + 10.0, A, H00, R0)
    assert abs(sn[1].max() - exact.max()) < 0.5
    assert rel_l2(sn[1], exact) < 3e-2
    assert abs(sn[1].sum() - H0.sum()) < 1e-12 * H0.sum()


def test_loss_weights_first_data_point_is_zero():
    """safe_slice rule (gradient.jl:38-40,144-149)."""
    ts = [0.0, 0.5, 1.0, 1.5]
    assert O.loss_weights(ts, [0.5, 1.0, 1.5]) == [0.0, 0.0, 0.5, 0.5]
    assert O.loss_weights(ts, ts) == [0.0, 0.5, 0.5, 0.5]
