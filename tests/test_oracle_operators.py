"""The reference's own unit identities for the staggered operators and their transposes
(test/SIA2D_adjoint_utils.jl:8-126): <u, A v> == <A^T u, v>, 10x11 randn, Delta = 2.5,
eta0 = 1, rtol 1e-11 -- run against the oracle's restatement."""
import numpy as np

from oracle import sia2d_oracle as O

SIZE = (10, 11)
FAC = SIZE[0] * SIZE[1]
RTOL = 1e-11


def _close(a, b):
    assert abs(a - b) <= RTOL * max(abs(a), abs(b))


def test_adjoint_diff():
    rng = np.random.default_rng(1234)
    for _ in range(5):
        u = rng.standard_normal(SIZE)
        v = rng.standard_normal((SIZE[0] - 1, SIZE[1]))
        _close(np.sum(O.diff_x(u) / 2.5 * v) / FAC, np.sum(u * O.diff_x_adjoint(v, 2.5)) / FAC)
    for _ in range(5):
        u = rng.standard_normal(SIZE)
        v = rng.standard_normal((SIZE[0], SIZE[1] - 1))
        _close(np.sum(O.diff_y(u) / 2.5 * v) / FAC, np.sum(u * O.diff_y_adjoint(v, 2.5)) / FAC)


def test_adjoint_avg():
    rng = np.random.default_rng(1234)
    for fwd, adj, shp in [(O.avg, O.avg_adjoint, (SIZE[0] - 1, SIZE[1] - 1)),
                          (O.avg_x, O.avg_x_adjoint, (SIZE[0] - 1, SIZE[1])),
                          (O.avg_y, O.avg_y_adjoint, (SIZE[0], SIZE[1] - 1))]:
        for _ in range(5):
            u = rng.standard_normal(SIZE)
            v = rng.standard_normal(shp)
            _close(np.sum(fwd(u) * v) / FAC, np.sum(u * adj(v)) / FAC)


def test_adjoint_clamp_borders():
    """test_adjoint_clamp_borders: <c, v> == <H, dH> + <dS, d_dS> (the clamp is piecewise
    linear and homogeneous of degree 1 in (dS, H))."""
    rng = np.random.default_rng(1234)
    D, eta0 = 2.5, 1.0
    for _ in range(5):
        H = np.abs(rng.standard_normal(SIZE))
        dS = rng.standard_normal((SIZE[0] - 1, SIZE[1] - 2))
        v = rng.standard_normal((SIZE[0] - 1, SIZE[1] - 2))
        c = O.clamp_borders_dx(dS, H, eta0, D)
        d_dS, d_H = O.clamp_borders_dx_adjoint(v, eta0, D, H, dS)
        _close(np.sum(c * v) / FAC, np.sum(H * d_H) / FAC + np.sum(dS * d_dS) / FAC)
    for _ in range(5):
        H = np.abs(rng.standard_normal(SIZE))
        dS = rng.standard_normal((SIZE[0] - 2, SIZE[1] - 1))
        v = rng.standard_normal((SIZE[0] - 2, SIZE[1] - 1))
        c = O.clamp_borders_dy(dS, H, eta0, D)
        d_dS, d_H = O.clamp_borders_dy_adjoint(v, eta0, D, H, dS)
        _close(np.sum(c * v) / FAC, np.sum(H * d_H) / FAC + np.sum(dS * d_dS) / FAC)


def test_l2sum_backward_vs_fd():
    """Manual backward of L2Sum (test/test_grad_loss.jl:405-447; 9x10 randn)."""
    rng = np.random.default_rng(1234)
    a, b = rng.standard_normal((9, 10)), rng.standard_normal((9, 10))
    mask = rng.random((9, 10)) > 0.3
    g = O.l2sum_backward(a, b, mask, 90.0)
    for (i, j) in [(0, 0), (3, 4), (8, 9), (5, 1)]:
        e = np.zeros_like(a)
        e[i, j] = 1e-6
        fd = (O.l2sum_loss(a + e, b, mask, 90.0) - O.l2sum_loss(a - e, b, mask, 90.0)) / 2e-6
        assert abs(fd - g[i, j]) <= 1e-8 * max(1.0, abs(fd))


def test_is_in_glacier_erosion():
    H = np.zeros((9, 9))
    H[1:8, 1:8] = 1.0
    m = O.is_in_glacier(H, 2)
    assert m.sum() == 9 and m[4, 4] and m[3, 3] and not m[2, 2]
    assert np.array_equal(O.is_in_glacier(H, 0), H > 0)
