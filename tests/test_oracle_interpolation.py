"""-m "not gpu": the oracle's restatement of create_interpolation (target_utils.jl:245-293) and of the `:Linear` branch
of dDiffusivity/dtheta for the Y law (target_D_hybrid.jl:136-160) -- the reference's DEFAULT for :D_hybrid."""
import numpy as np

from oracle import sia2d_oracle as O


def test_create_interpolation_knots():
    rng = np.random.default_rng(0)
    A = np.abs(rng.standard_normal((40, 31))) * 80.0
    A[A < 15.0] = 0.0  # ice-free nodes
    n = 20
    k = O.create_interpolation(A, n)
    assert k.size == 2 * n and np.all(np.diff(k) > 0)
    assert k[0] == 0.0 and k[-1] == A.max()
    # the uniform half is LinRange(0, max, n); the other half are type-7 quantiles of the entries in (0, max)
    unif = np.arange(n) / (n - 1.0) * A.max()
    assert all(np.isclose(k, u, rtol=0, atol=1e-12).any() for u in unif)
    inside = A[(A > 0) & (A < A.max())]
    q = np.quantile(inside, (np.arange(n + 2) / (n + 1.0))[1:-1])  # numpy's default "linear" method == type 7
    assert all(np.isclose(k, v, rtol=1e-14, atol=0).any() for v in q)
    # keyword variants used by feed_input_cache! (src/laws/Cache.jl:142-153)
    k2 = O.create_interpolation(A, n, dilation_factor=1.05, minA_quantile=10.0)
    assert np.isclose(k2[-1], 1.05 * A.max()) and k2[0] == 0.0


def test_linear_interpolation_is_exact_for_affine_knot_values():
    nodes = np.array([0.0, 1.0, 2.5, 7.0])
    x = np.array([0.0, 0.3, 1.0, 2.0, 6.9, 7.0])
    k, w = O.interp_linear_weights(nodes, x)
    v = 3.0 * nodes - 2.0
    assert np.allclose((1 - w) * v[k] + w * v[k + 1], 3.0 * x - 2.0, rtol=0, atol=1e-14)
    assert k.max() <= len(nodes) - 2 and np.all((w >= 0) & (w <= 1))


def test_Y_law_gradient_linear_vs_exact():
    """The interpolated gradient converges to the exact per-node one as the knots get denser, and the default
    (n_interp_half = 75) is already within a few 1e-4 -- "probably sufficient" in the reference's words."""
    ph = O.Phys()
    H, B = O.synthetic_valley(48, 40, 50.0)
    mlp = O.MLP([2, 3, 10, 3, 1], [1, 1, 1, 2], [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
    th = mlp.init_theta(np.random.default_rng(9)) + 0.05 * np.random.default_rng(10).standard_normal(mlp.n_params)
    lam = np.random.default_rng(3).standard_normal(H.shape)
    exact = O.vjp_theta(lam, H, B, 50.0, 50.0, ph, O.Law(kind=O.LAW_NN_Y, mlp=mlp, theta=th, T=-5.0, interpolation="none"))
    err = []
    for n in (10, 75, 600):
        law = O.Law(kind=O.LAW_NN_Y, mlp=mlp, theta=th, T=-5.0, interpolation="linear", n_interp_half=n)
        g = O.vjp_theta(lam, H, B, 50.0, 50.0, ph, law)
        err.append(np.linalg.norm(g - exact) / np.linalg.norm(exact))
    assert err[0] > err[1] > err[2] and err[1] < 2e-3 and err[2] < 1e-4, err
    # the reference's default for the Y law IS the linear branch with 75 knots per half
    dflt = O.vjp_theta(lam, H, B, 50.0, 50.0, ph, O.Law(kind=O.LAW_NN_Y, mlp=mlp, theta=th, T=-5.0))
    l75 = O.vjp_theta(lam, H, B, 50.0, 50.0, ph, O.Law(kind=O.LAW_NN_Y, mlp=mlp, theta=th, T=-5.0, interpolation="linear", n_interp_half=75))
    assert np.array_equal(dflt, l75)
    assert O.Law(kind=O.LAW_NN_U).interp()[0] == "none" and O.Law(kind=O.LAW_NN_A_SCALAR).interp()[0] == "none"


def test_U_law_gradient_bilinear_node_grid():
    """SIA2D_D_target(interpolation = :Linear) (target_D_pure.jl:179-193, Laws.jl:128-169): gradients on a fixed node grid
    over [0, 100]^2, bilinear in (Hbar, |grad S|).  Checked: the bilinear weights reproduce a function affine in both
    arguments exactly; with the slope axis the reference really uses ([0, 100]: every physical slope sits in the first
    interval) the branch differs from the exact one by the interpolation error, which shrinks with the node count;
    a node with Hbar > 100 raises (the gridded interpolant throws); the law's default stays :None."""
    import pytest
    ph = O.Phys()
    H, B = O.synthetic_icecap(48, 40, 100.0)
    H = H * (80.0 / H.max())
    mlp = O.MLP([2, 3, 10, 3, 1], [1, 1, 1, 2], [(0.0, 300.0), (0.0, 0.5)], O.POST_EXPMAX, 0.0, 50.0)
    th = mlp.init_theta(np.random.default_rng(9)) + 0.05 * np.random.default_rng(10).standard_normal(mlp.n_params)
    lam = np.random.default_rng(3).standard_normal(H.shape)
    exact = O.vjp_theta(lam, H, B, 100.0, 100.0, ph, O.Law(kind=O.LAW_NN_U, mlp=mlp, theta=th))
    err = []
    for n in (10, 100, 250):
        law = O.Law(kind=O.LAW_NN_U, mlp=mlp, theta=th, interpolation="linear", n_interp_half=n)
        g = O.vjp_theta(lam, H, B, 100.0, 100.0, ph, law)
        err.append(np.linalg.norm(g - exact) / np.linalg.norm(exact))
    assert err[0] > err[1] > err[2] and np.isfinite(err).all(), err
    # direct evaluation of the definition on one node: sum of the four corner gradients with tent weights
    Hc = np.maximum(H, 0.0)
    Hbar = O.avg(Hc)
    S = B + Hc
    gS = np.sqrt(O.avg_y(O.diff_x(S) / 100.0) ** 2 + O.avg_x(O.diff_y(S) / 100.0) ** 2)
    G = O.law_grad_theta_bilinear(O.Law(kind=O.LAW_NN_U, mlp=mlp, theta=th), ph, Hbar, gS, th, 10)
    K = 20
    nodes = np.arange(K) / (K - 1.0) * 100.0
    i, j = np.unravel_index(np.argmax(Hbar), Hbar.shape)
    kh = int(np.searchsorted(nodes, Hbar[i, j], side="right") - 1)
    wh = (Hbar[i, j] - nodes[kh]) / (nodes[kh + 1] - nodes[kh])
    ws = gS[i, j] / nodes[1]
    corner = lambda a, b_: O.mlp_grad_theta(mlp, th, np.array([[nodes[a]], [nodes[b_]]]))[:, 0]
    want = (1 - wh) * (1 - ws) * corner(kh, 0) + wh * (1 - ws) * corner(kh + 1, 0) + (1 - wh) * ws * corner(kh, 1) \
        + wh * ws * corner(kh + 1, 1)
    assert np.allclose(G[:, i, j], want, rtol=1e-12, atol=1e-300)
    with pytest.raises(IndexError):
        O.vjp_theta(lam, H * 2.0, B, 100.0, 100.0, ph, O.Law(kind=O.LAW_NN_U, mlp=mlp, theta=th, interpolation="linear"))
    assert O.Law(kind=O.LAW_NN_U).interp() == ("none", 100)
