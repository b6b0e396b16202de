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


def test_laplacian_transpose_and_tikhonov_gradient():
    """test_grad_TikhonovRegularization (test/test_grad_loss.jl:449-487): 9x10 field, dx=1.2, dy=1.8,
    random mask; backward_loss == exact gradient (the loss is quadratic, so central differences are
    exact up to rounding) and <lap a, lam> == <a, VJP lam>."""
    rng = np.random.default_rng(3)
    nx, ny, dx, dy = 9, 10, 1.2, 1.8
    a = rng.standard_normal((nx, ny))
    a[:2, :] = 0; a[-2:, :] = 0; a[:, :2] = 0; a[:, -2:] = 0
    lam = rng.standard_normal((nx, ny))
    assert abs(np.sum(O.laplacian(a, dx, dy) * lam) - np.sum(a * O.vjp_laplacian(lam, dx, dy))) < 1e-12
    mask = rng.standard_normal((nx, ny)) >= 0
    g = O.tikhonov_backward(a, dx, dy, mask)
    gn = np.zeros_like(a)
    for i in range(nx):
        for j in range(ny):
            e = np.zeros_like(a)
            e[i, j] = 1e-3
            gn[i, j] = (O.tikhonov_loss(a + e, dx, dy, mask) - O.tikhonov_loss(a - e, dx, dy, mask)) / 2e-3
    assert np.abs(g - gn).max() <= 1e-11 * np.abs(g).max()
    # boundary ring of the Laplacian is zero
    L = O.laplacian(a, dx, dy)
    assert not L[0, :].any() and not L[-1, :].any() and not L[:, 0].any() and not L[:, -1].any()


def test_initial_condition_filters():
    """evaluate_H0 / evaluate_dH0 (InitialCondition_utils.jl:30-141): derivative == d(filter)/dx, mask zeroed."""
    rng = np.random.default_rng(5)
    x = rng.uniform(-3, 3, (7, 6))
    outside = rng.uniform(size=x.shape) < 0.3
    for f in ("identity", "softplus", "Zang1980"):
        H = O.evaluate_H0(x, outside, f)
        d = O.evaluate_dH0(x, outside, f)
        e = 1e-6
        fd = (O.evaluate_H0(x + e, outside, f) - O.evaluate_H0(x - e, outside, f)) / (2 * e)
        assert np.abs(fd - d).max() < 1e-6
        assert not H[outside].any() and not d[outside].any()
    assert O.sigma_zang(np.array([-1.0]))[0] == 0.0 and O.sigma_zang(np.array([1.0]))[0] == 1.0  # C1 joins
    assert abs(O.sigma_zang(np.array([0.0]))[0] - 0.25) < 1e-15
