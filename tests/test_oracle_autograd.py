"""-m "not gpu": an AD-exact pin of the oracle's hand-written adjoints.

The reference checks its hand VJPs and loss pull-backs against Enzyme to 1e-14 (test/test_grad_loss.jl:405-496,
test/SIA2D_adjoint.jl:2-207).  Julia / Enzyme are not in this image, torch is: the FORWARD operators (Huginn.SIA2D!,
surface_V, the mass-balance step, the loss terms, the initial-condition filters) are re-expressed here in torch fp64 from
SURVEY Appendix A -- slices and elementwise operations written independently of oracle/sia2d_oracle.py's helpers -- and
`torch.autograd` VJPs are compared with oracle.vjp_H / vjp_theta / vjp_surface_V_* / vjp_mb / *_backward at 1e-12.
Until now the oracle's adjoints were pinned by finite differences only (5e-7 ... 5e-3); oracle and kernels were written by
the same hand from the same text, so a common-mode error in an adjoint would have been FD-visible only.

Where the hand adjoint is NOT the derivative of the forward (all listed, each with its own assertion below):
  * exact ties of the slope clamp: clamp_borders_dx/dy_adjoint use strict inequalities (inversion_utils.jl:22-43), so at
    dS == bound the whole contribution is dropped, AD splits it between the two arguments of min / max;
  * H == 0: the final mask `H .> 0` (adjoint.jl:148) is d max(H, 0)/dH with the convention 0 at 0 (torch.relu's);
  * VJP_lambda_dsurface_V/dH (adjoint.jl:268-350) clamps H like SIA2D! but has NO such mask: on cells with H <= 0 it returns
    the derivative with respect to the clamped thickness where AD returns 0 (so do backward_loss(::LossV) and the
    velocity regulariser, which call it); exact on every cell with H > 0;
  * target :D's beta = dD/d|grad S| is NOT divided by |grad S| (target_D_pure.jl:123-137, "for now we ignore the derivative
    in surface slope") although adjoint.jl:122-127 multiplies it by the slope components: the U law's H-VJP as written is
    not the transpose of its RHS (its alpha is);
  * the Y and U laws' alpha / beta are finite differences of the law by construction (target_D_hybrid.jl:58-71,
    target_D_pure.jl:105-137): the ASSEMBLY is checked exactly by substituting AD partials, the differences to their
    truncation error;
  * surface velocity with sliding (C != 0): dVelocity^/dH as written (target_A.jl:110-125) lacks the factor (p - q + 1);
    LossV(component = :abs) distributes dl/dV as (V_x - V_x_ref) / (V - V_ref) (Losses.jl:367-368), not V_x / V;
    target :D's slope partial is not divided by |grad S| (target_D_pure.jl:233-245) although the H-VJP multiplies by the
    slope component; target :D_hybrid mixes Gamma and Gamma^ (target_D_hybrid.jl:210-372).  These are restated as written
    and are exercised by the FD tests at the reference's own tolerances, not here.
"""
import math

import numpy as np
import pytest

from oracle import sia2d_oracle as O

torch = pytest.importorskip("torch")
torch.set_default_dtype(torch.float64)
TOL = 1e-12


def T(a, grad=False):
    t = torch.tensor(np.asarray(a, dtype=np.float64))
    return t.requires_grad_(True) if grad else t


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (d if d > 0 else 1.0))


# ---- torch restatement of the forward operators ------------------------------------------------------

def t_act(code, x):
    if code == O.ACT_IDENTITY:
        return x
    if code == O.ACT_SOFTPLUS:
        return torch.log1p(torch.exp(-torch.abs(x))) + torch.relu(x)
    if code == O.ACT_SIGMOID:
        return torch.sigmoid(x)
    if code == O.ACT_GELU:
        return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))
    if code == O.ACT_TANH:
        return torch.tanh(x)
    if code == O.ACT_RELU:
        return torch.relu(x)
    raise ValueError(code)


def t_mlp(mlp, theta, xs):
    """xs: list of n_in tensors of one shape; Lux Dense layers, theta = [vec(W) column-major, b] per layer."""
    h = list(xs)
    if mlp.prescale is not None:
        h = [(x - lo) / (hi - lo) - 0.5 for x, (lo, hi) in zip(h, mlp.prescale)]
    o = 0
    for l, a in enumerate(mlp.acts):
        nin, nout = mlp.widths[l], mlp.widths[l + 1]
        W = theta[o:o + nin * nout].reshape(nin, nout)  # column-major (nout x nin): element [r, c] at c * nout + r
        o += nin * nout
        b = theta[o:o + nout]
        o += nout
        h = [t_act(a, sum(W[c, r] * h[c] for c in range(nin)) + b[r]) for r in range(nout)]
    y = h[0]
    if mlp.post_kind == O.POST_AFFINE:
        y = mlp.post_lo + (mlp.post_hi - mlp.post_lo) * y
    elif mlp.post_kind == O.POST_EXPMAX:
        y = mlp.post_hi * torch.exp((y - 1.0) / y)
    elif mlp.post_kind == O.POST_SCALE:
        y = mlp.post_hi * y
    return y


def t_geometry(H, B, dx, dy):
    Hc = torch.relu(H)
    S = B + Hc
    sx = (S[1:, :] - S[:-1, :]) / dx      # (nx-1, ny)
    sy = (S[:, 1:] - S[:, :-1]) / dy      # (nx, ny-1)
    gx = 0.5 * (sx[:, :-1] + sx[:, 1:])   # (nx-1, ny-1)
    gy = 0.5 * (sy[:-1, :] + sy[1:, :])
    gS = torch.sqrt(gx * gx + gy * gy)
    Hb = 0.25 * (Hc[:-1, :-1] + Hc[1:, :-1] + Hc[:-1, 1:] + Hc[1:, 1:])
    return Hc, S, sx, sy, gx, gy, gS, Hb


def t_law_value(law, ph, theta, Hb, gS):
    if law.kind == O.LAW_CONST_A:
        return theta  # "theta" is A itself (scalar tensor or dual field)
    if law.kind == O.LAW_NN_A_SCALAR:
        return t_mlp(law.mlp, theta, [T(float(law.T))])
    if law.kind == O.LAW_NN_A_GRIDDED:
        return t_mlp(law.mlp, theta, [T(law.T)])
    if law.kind == O.LAW_NN_Y:
        return t_mlp(law.mlp, theta, [torch.full_like(Hb, float(law.T)), Hb])
    return t_mlp(law.mlp, theta, [Hb, gS])


def t_diffusivity(law, ph, theta, Hb, gS):
    val = t_law_value(law, ph, theta, Hb, gS)
    if law.kind == O.LAW_NN_U:
        return Hb * val
    nH, nS = (law.n_H or ph.n, law.n_gradS or ph.n) if law.kind == O.LAW_NN_Y else (ph.n, ph.n)
    D = val * (2.0 * (ph.rho * ph.g) ** ph.n / (ph.n + 2.0)) * Hb ** (nH + 2.0) * gS ** (nS - 1.0)
    if ph.C != 0.0:
        D = D + ph.C * (ph.rho * ph.g) ** (ph.p - ph.q) * Hb ** (ph.p - ph.q + 1.0) * gS ** (ph.p - 1.0)
    return D


def t_rhs(H, B, dx, dy, ph, law, theta):
    Hc, S, sx, sy, gx, gy, gS, Hb = t_geometry(H, B, dx, dy)
    D = t_diffusivity(law, ph, theta, Hb, gS)
    ex, ey = sx[:, 1:-1], sy[1:-1, :]
    exc = torch.maximum(torch.minimum(ex, ph.eta0 * Hc[1:, 1:-1] / dx), -ph.eta0 * Hc[:-1, 1:-1] / dx)
    eyc = torch.maximum(torch.minimum(ey, ph.eta0 * Hc[1:-1, 1:] / dy), -ph.eta0 * Hc[1:-1, :-1] / dy)
    Fx = -0.5 * (D[:, :-1] + D[:, 1:]) * exc     # (nx-1, ny-2)
    Fy = -0.5 * (D[:-1, :] + D[1:, :]) * eyc     # (nx-2, ny-1)
    inner = -((Fx[1:, :] - Fx[:-1, :]) / dx + (Fy[:, 1:] - Fy[:, :-1]) / dy)
    return torch.nn.functional.pad(inner, (1, 1, 1, 1))


def t_surface_V(H, B, dx, dy, ph, law, theta):
    Hc, S, sx, sy, gx, gy, gS, Hb = t_geometry(H, B, dx, dy)
    if law.kind == O.LAW_NN_U:
        Vup = t_law_value(law, ph, theta, Hb, gS) / law.fV
    else:
        Vup = t_law_value(law, ph, theta, Hb, gS) * (2.0 * (ph.rho * ph.g) ** ph.n / (ph.n + 1.0)) * Hb ** (ph.n + 1.0) * gS ** (ph.n - 1.0)
    return -Vup * gx, -Vup * gy


# ---- inputs ------------------------------------------------------------------------------------------------------------------

def fields(seed, nx=13, ny=11, dx=40.0, holes=True, hmax=120.0):
    """random bed (no two equal slopes: no clamp ties), thickness with an ice-free corner (H == 0 exactly) and a few negative
    cells (Huginn.SIA2D! clamps them, adjoint.jl:52)"""
    rng = np.random.default_rng(seed)
    x = np.arange(nx)[:, None] * dx
    B = 1500.0 - 0.05 * x + 8.0 * rng.standard_normal((nx, ny))
    H = hmax * (0.3 + 0.7 * rng.random((nx, ny)))
    if holes:
        H[: nx // 3, : ny // 3] = 0.0
        H[nx // 2, ny - 2] = -3.0
        H[1, ny - 3] = -0.5
    return H, B, rng


def laws(rng, which):
    ph = O.Phys()
    if which == "constA":
        return ph, O.Law(kind=O.LAW_CONST_A, A=3.1e-17), None
    if which == "constA_sliding_n32":
        return O.Phys(n=3.2, C=7e-8, p=3.0, q=1.0), O.Law(kind=O.LAW_CONST_A, A=3.1e-17), None
    if which == "nnA":
        m = O.default_nn(1, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
        th = m.init_theta(rng) + 0.1 * rng.standard_normal(m.n_params)
        return ph, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=m, theta=th, T=-7.5), th
    if which == "nnA_gridded":
        m = O.MLP([1, 5, 4, 1], [O.ACT_GELU, O.ACT_TANH, O.ACT_SIGMOID], post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
        th = m.init_theta(rng) + 0.1 * rng.standard_normal(m.n_params)
        return ph, O.Law(kind=O.LAW_NN_A_GRIDDED, mlp=m, theta=th, T=None), th
    if which == "Y":
        m = O.default_nn(2, prescale=[(-25.0, 0.0), (0.0, 500.0)], post_kind=O.POST_EXPMAX, post_hi=ph.maxA)
        th = m.init_theta(rng) + 0.1 * rng.standard_normal(m.n_params)
        return ph, O.Law(kind=O.LAW_NN_Y, mlp=m, theta=th, T=-4.0, interpolation="none"), th
    if which == "U":
        m = O.MLP([2, 4, 3, 1], [O.ACT_SOFTPLUS, O.ACT_GELU, O.ACT_SIGMOID], prescale=[(0.0, 300.0), (0.0, 0.5)],
                  post_kind=O.POST_EXPMAX, post_hi=50.0)
        th = m.init_theta(rng) + 0.1 * rng.standard_normal(m.n_params)
        return ph, O.Law(kind=O.LAW_NN_U, mlp=m, theta=th, fV=0.8), th
    raise ValueError(which)


def torch_theta(law, th):
    return T(law.A, grad=True) if law.kind == O.LAW_CONST_A else T(th, grad=True)


# ---- SIA2D!: J_H^T lambda and J_theta^T lambda -------------------------------------------------------------------------------

@pytest.mark.parametrize("which", ["constA", "constA_sliding_n32", "nnA", "nnA_gridded"])
def test_vjp_H_and_theta_equal_autograd_of_the_rhs(which):
    """test_adjoint_SIA2D against Enzyme (test/SIA2D_adjoint.jl:139-206) with torch in Enzyme's place: closed-form laws."""
    H, B, rng = fields(11)
    ph, law, th = laws(rng, which)
    if which == "nnA_gridded":
        law.T = -12.0 + 10.0 * rng.random((H.shape[0] - 1, H.shape[1] - 1))
    lam = rng.standard_normal(H.shape)
    Ht, tht = T(H, grad=True), torch_theta(law, th)
    f = t_rhs(Ht, T(B), 40.0, 40.0, ph, law, tht)
    assert rel(f.detach().numpy(), O.sia2d_rhs(H, B, 40.0, 40.0, ph, law, th)) < TOL
    gH, gth = torch.autograd.grad((f * T(lam)).sum(), [Ht, tht])
    assert rel(O.vjp_H(lam, H, B, 40.0, 40.0, ph, law, th), gH.numpy()) < TOL
    assert rel(O.vjp_theta(lam, H, B, 40.0, 40.0, ph, law, th), gth.numpy().reshape(-1)) < TOL
    assert np.all(O.vjp_H(lam, H, B, 40.0, 40.0, ph, law, th)[H <= 0.0] == 0.0)  # the H > 0 mask == relu'(H), 0 at 0


@pytest.mark.parametrize("which", ["Y", "U"])
def test_vjp_H_assembly_is_exact_with_autograd_partials_and_theta_vjp_is_exact(which, monkeypatch):
    """Per-node-MLP laws: the reference forms alpha = dD/dHbar and beta = (dD/d|grad S|) / |grad S| by finite differences of
    the law (1e-4 forward for Y's network part, central 1e-4 / 1e-6 for U).  (i) With alpha, beta replaced by the AD
    partials of D(Hbar, |grad S|), the oracle's assembly (adjoint.jl:99-148) equals autograd of the RHS to 1e-12;
    (ii) the finite-difference partials agree with the AD partials to their truncation error; (iii) the theta-VJP
    (exact backprop per node, :None) equals autograd to 1e-12."""
    H, B, rng = fields(12, holes=(which == "Y"))  # (U law: dD/dtheta is masked by Hbar > 0, and D = Hbar U vanishes there anyway)
    ph, law, th = laws(rng, which)
    lam = rng.standard_normal(H.shape)
    Ht, tht = T(H, grad=True), T(th, grad=True)
    f = t_rhs(Ht, T(B), 40.0, 40.0, ph, law, tht)
    assert rel(f.detach().numpy(), O.sia2d_rhs(H, B, 40.0, 40.0, ph, law, th)) < TOL
    gH, gth = torch.autograd.grad((f * T(lam)).sum(), [Ht, tht])
    assert rel(O.vjp_theta(lam, H, B, 40.0, 40.0, ph, law, th), gth.numpy()) < TOL

    def ad_partials(law_, ph_, Hbar, gradS, theta=None):
        hb, gs = T(Hbar, grad=True), T(gradS, grad=True)
        D = t_diffusivity(law_, ph_, T(th), hb, gs)
        a, b = torch.autograd.grad(D.sum(), [hb, gs])
        return a.numpy(), b.numpy() / gradS

    fd_alpha, fd_beta = O.d_diffusivity_dH, O.d_diffusivity_dgradS
    monkeypatch.setattr(O, "d_diffusivity_dH", lambda *a, **k: ad_partials(*a, **k)[0])
    monkeypatch.setattr(O, "d_diffusivity_dgradS", lambda *a, **k: ad_partials(*a, **k)[1])
    assert rel(O.vjp_H(lam, H, B, 40.0, 40.0, ph, law, th), gH.numpy()) < TOL
    monkeypatch.undo()
    # the reference's finite differences against the exact partials (truncation: 1e-4 forward / central differences)
    _, _, _, _, gS, Hbar, *_ = O._forward_intermediates(H, B, 40.0, 40.0, ph)
    a_ad, b_ad = ad_partials(law, ph, Hbar, gS)
    ice = Hbar > 0.0
    assert rel(fd_alpha(law, ph, Hbar, gS, th)[ice], a_ad[ice]) < 1e-5
    hand = O.vjp_H(lam, H, B, 40.0, 40.0, ph, law, th)
    if which == "Y":
        assert rel(fd_beta(law, ph, Hbar, gS, th)[ice], b_ad[ice]) < 1e-5
        assert rel(hand, gH.numpy()) < 1e-5  # as written: FD-accurate (2.6e-11 here: the network part of alpha is small)
    else:
        # target :D as written: beta is dD/d|grad S| itself (target_D_pure.jl:123-137), not divided by |grad S| ...
        assert rel(fd_beta(law, ph, Hbar, gS, th)[ice], (b_ad * gS)[ice]) < 1e-5
        # ... so its H-VJP is not the transpose of the RHS (the slope term is off by the factor |grad S| ~ 0.2 here)
        assert rel(hand, gH.numpy()) > 1e-3


def test_clamp_ties_are_where_the_hand_adjoint_and_autograd_differ():
    """Flat bed, a margin cell next to an ice-free cell: S2 - S1 == H bit for bit, the clamp sits on its bound.  The
    reference's adjoint (strict inequalities, inversion_utils.jl:22-43) drops the edge's contribution -- neither
    d(dS) nor dH gets it -- while AD of min / max splits it 1/2 : 1/2.  The two differ ONLY on the cells of tied edges."""
    nx, ny, dx = 12, 10, 50.0
    rng = np.random.default_rng(5)
    B = np.full((nx, ny), 1000.0)
    H = np.zeros((nx, ny))
    H[3:9, 2:8] = 64.0 + 8.0 * rng.integers(0, 4, (6, 6))  # lattice-valued plateau: exact differences
    ph, law = O.Phys(), O.Law(kind=O.LAW_CONST_A, A=3.1e-17)
    lam = rng.standard_normal((nx, ny))
    Ht = T(H, grad=True)
    f = t_rhs(Ht, T(B), dx, dx, ph, law, T(law.A))
    (gH,) = torch.autograd.grad((f * T(lam)).sum(), [Ht])
    hand = O.vjp_H(lam, H, B, dx, dx, ph, law)
    S = B + H
    ex, ey = np.diff(S[:, 1:-1], axis=0) / dx, np.diff(S[1:-1, :], axis=1) / dx
    tie_x = (ex == ph.eta0 * H[1:, 1:-1] / dx) | (ex == -ph.eta0 * H[:-1, 1:-1] / dx)
    tie_y = (ey == ph.eta0 * H[1:-1, 1:] / dx) | (ey == -ph.eta0 * H[1:-1, :-1] / dx)
    near = np.zeros((nx, ny), bool)
    near[:-1, 1:-1] |= tie_x; near[1:, 1:-1] |= tie_x
    near[1:-1, :-1] |= tie_y; near[1:-1, 1:] |= tie_y
    assert tie_x.sum() + tie_y.sum() > 10
    diff = np.abs(hand - gH.numpy()) > 1e-12 * np.abs(hand).max()
    assert diff.any() and not (diff & ~near).any()


# ---- surface velocity (target :A, C = 0: the partials as written are the derivatives) ------------------------------------

@pytest.mark.parametrize("which", ["constA", "nnA", "nnA_gridded"])
def test_surface_V_vjps_equal_autograd(which):
    """test_adjoint_surface_V (test/SIA2D_adjoint.jl:209-216) with AD instead of FD."""
    H, B, rng = fields(21)
    ph, law, th = laws(rng, which)
    if which == "nnA_gridded":
        law.T = -12.0 + 10.0 * rng.random((H.shape[0] - 1, H.shape[1] - 1))
    dVx, dVy = rng.standard_normal(H.shape), rng.standard_normal(H.shape)
    Ht, tht = T(H, grad=True), torch_theta(law, th)
    vx, vy = t_surface_V(Ht, T(B), 40.0, 40.0, ph, law, tht)
    ox, oy = O.surface_V(H, B, 40.0, 40.0, ph, law, th)
    assert rel(vx.detach().numpy(), ox) < TOL and rel(vy.detach().numpy(), oy) < TOL
    gH, gth = torch.autograd.grad((vx * T(dVx[:-1, :-1])).sum() + (vy * T(dVy[:-1, :-1])).sum(), [Ht, tht])
    hand, ice = O.vjp_surface_V_H(dVx, dVy, H, B, 40.0, 40.0, ph, law, th), H > 0.0
    assert rel(hand[ice], gH.numpy()[ice]) < TOL
    # (no H > 0 mask in adjoint.jl:268-350: where H <= 0 the hand VJP is the derivative w.r.t. the clamped thickness, AD's is 0)
    assert np.all(gH.numpy()[~ice] == 0.0) and np.abs(hand[~ice]).max() > 0.0
    assert rel(O.vjp_surface_V_theta(dVx, dVy, H, B, 40.0, 40.0, ph, law, th), gth.numpy().reshape(-1)) < TOL


def test_surface_V_theta_vjp_of_the_U_law_equals_autograd_on_an_ice_covered_domain():
    """dVelocity^/dtheta = (Hbar > 0) dU/dtheta / f (target_D_pure.jl:139-176,247-255): exact where every node carries ice."""
    H, B, rng = fields(22, holes=False)
    ph, law, th = laws(rng, "U")
    dVx, dVy = rng.standard_normal(H.shape), rng.standard_normal(H.shape)
    tht = T(th, grad=True)
    vx, vy = t_surface_V(T(H), T(B), 40.0, 40.0, ph, law, tht)
    (gth,) = torch.autograd.grad((vx * T(dVx[:-1, :-1])).sum() + (vy * T(dVy[:-1, :-1])).sum(), [tht])
    assert rel(O.vjp_surface_V_theta(dVx, dVy, H, B, 40.0, 40.0, ph, law, th), gth.numpy()) < TOL


def test_lossV_xy_backward_equals_autograd():
    """backward_loss(::LossV, component = :xy, scale_loss) (Losses.jl:338-390): dL/dH and dL/dtheta."""
    H, B, rng = fields(23)
    ph, law, th = laws(rng, "nnA")
    Vx, Vy, V = O.V_from_H(H * 1.07, B, 40.0, 40.0, ph, law, th)
    Vabs_ref = V.copy()
    Vabs_ref[2:4, 5:7] = 0.0  # masked-out data
    spec = O.LossVSpec(component="xy", scale_loss=True)
    N = float(H.size)
    Ht, tht = T(H, grad=True), T(th, grad=True)
    vx, vy = t_surface_V(Ht, T(B), 40.0, 40.0, ph, law, tht)
    m = T((Vabs_ref > 0.0)[:-1, :-1].astype(float))
    sc = 1.0 / math.sqrt(np.mean(Vx[Vabs_ref > 0.0] ** 2 + Vy[Vabs_ref > 0.0] ** 2))
    L = ((m * (vx - T(Vx[:-1, :-1])) ** 2).sum() + (m * (vy - T(Vy[:-1, :-1])) ** 2).sum()) / N * sc
    assert abs(L.item() - O.loss_V(spec, H, B, 40.0, 40.0, ph, law, Vabs_ref, Vx, Vy, N, th)) < TOL * abs(L.item())
    gH, gth = torch.autograd.grad(L, [Ht, tht])
    dH, dth = O.backward_loss_V(spec, H, B, 40.0, 40.0, ph, law, Vabs_ref, Vx, Vy, N, th)
    ice = H > 0.0
    assert rel(dH[ice], gH.numpy()[ice]) < TOL and rel(dth, gth.numpy()) < TOL  # (H <= 0: see test_surface_V_vjps_equal_autograd)


# ---- mass balance -------------------------------------------------------------------------------------------------------------

def test_mass_balance_vjp_equals_autograd_of_the_applied_increment():
    """VJP_lambda_dMB/dH (VJPs.jl:107-151): the applied increment MB(H) -- masked (:129-131), clipped where the ice would
    disappear (:133-139), elevation feedback saturating at mb_max -- has a diagonal Jacobian; AD of the same expression."""
    rng = np.random.default_rng(31)
    nx, ny = 12, 9
    B = 2000.0 + 30.0 * rng.standard_normal((nx, ny))
    H = 25.0 * rng.random((nx, ny))
    H[:3, :] = 0.0
    S_ref = B + 10.0
    mb = O.MassBalance(mb0=-6.0 + 9.0 * rng.random((nx, ny)), dmb_dS=0.35, S_ref=S_ref, mb_max=1.5)
    lam = rng.standard_normal((nx, ny))
    Ht = T(H, grad=True)
    raw = T(mb.mb0) + mb.dmb_dS * ((T(B) + Ht) - T(S_ref))
    MB = torch.where(raw >= mb.mb_max, torch.full_like(raw, mb.mb_max), raw)
    mask = ((Ht > 0.0) & (MB < 0.0)) | ((Ht > 10.0) & (MB >= 0.0))
    MB = torch.where(mask, MB, torch.zeros_like(MB))
    MB = torch.where(mask & ((Ht + MB) < 0.0), -Ht, MB)
    Hn, MBo = O.mb_apply(mb, H, B)
    assert rel(MB.detach().numpy(), MBo) < TOL and rel((Ht + MB).detach().numpy(), Hn) < TOL
    cases = (mask.numpy().sum(), (~mask.numpy()).sum(), (MBo == -H)[mask.numpy()].sum(), (raw.detach().numpy() >= mb.mb_max).sum())
    assert all(c > 0 for c in cases), cases  # every branch is exercised
    (g,) = torch.autograd.grad((MB * T(lam)).sum(), [Ht])
    assert rel(O.vjp_mb(mb, lam, H, B), g.numpy()) < TOL


# ---- simple losses, Tikhonov, regularisers, time-aggregated terms, IC filters --------------------------------------------------

def test_simple_loss_backwards_equal_autograd():
    """test_grad_L2Sum (test/test_grad_loss.jl:405-442) and the LogSum analogue (Losses.jl:207-229)."""
    rng = np.random.default_rng(41)
    a, b = 50.0 * rng.random((9, 10)), 50.0 * rng.random((9, 10))
    mask = O.is_in_glacier(np.pad(np.ones((7, 8)), 1), 1) | (rng.random((9, 10)) > 0.6)
    at = T(a, grad=True)
    m = T(mask.astype(float))
    (g,) = torch.autograd.grad((m * (at - T(b)) ** 2).sum() / 3.5, [at])
    assert rel(O.l2sum_backward(a, b, mask, 3.5), g.numpy()) < 1e-14
    (g,) = torch.autograd.grad((m * torch.log((at + 0.1) / (T(b) + 0.1)) ** 2).sum() / 3.5, [at])
    assert rel(O.logsum_backward(a, b, mask, 3.5, 0.1), g.numpy()) < 1e-14
    assert abs(O.logsum_loss(a, b, mask, 3.5, 0.1) - (m * torch.log((at + 0.1) / (T(b) + 0.1)) ** 2).sum().item() / 3.5) < 1e-12


def t_laplacian(a, dx, dy):
    """the reference's staggered Laplacian (Regularization.jl:330-352) as ONE 3x3 stencil, derived independently:
    [1 2 1]^T/4 (x) dxx/dx^2 + dyy/dy^2 (x) [1 2 1]/4 on the interior, 0 on the ring"""
    c = a[1:-1, 1:-1]
    dxx = lambda j0, j1: (a[2:, j0:j1] - 2.0 * a[1:-1, j0:j1] + a[:-2, j0:j1]) / (dx * dx)
    dyy = lambda i0, i1: (a[i0:i1, 2:] - 2.0 * a[i0:i1, 1:-1] + a[i0:i1, :-2]) / (dy * dy)
    ny, nx = a.shape[1], a.shape[0]
    lx = 0.25 * (dxx(0, ny - 2) + 2.0 * dxx(1, ny - 1) + dxx(2, ny))
    ly = 0.25 * (dyy(0, nx - 2) + 2.0 * dyy(1, nx - 1) + dyy(2, nx))
    return torch.nn.functional.pad(lx + ly + 0.0 * c, (1, 1, 1, 1))


def test_tikhonov_backward_equals_autograd():
    """test_grad_TikhonovRegularization (test/test_grad_loss.jl:444-496): dx != dy, random mask."""
    rng = np.random.default_rng(42)
    a = rng.standard_normal((9, 10))
    mask = rng.standard_normal((9, 10)) >= 0
    at = T(a, grad=True)
    lap = t_laplacian(at, 1.2, 1.8)
    assert rel(lap.detach().numpy(), O.laplacian(a, 1.2, 1.8)) < 1e-13
    L = (T(mask.astype(float)) * lap ** 2).sum()
    assert abs(L.item() - O.tikhonov_loss(a, 1.2, 1.8, mask)) < 1e-12 * abs(L.item())
    (g,) = torch.autograd.grad(L, [at])
    assert rel(O.tikhonov_backward(a, 1.2, 1.8, mask), g.numpy()) < 1e-13


def test_velocity_regularization_backward_equals_autograd():
    """VelocityRegularization (Regularization.jl:192-245): sum_mask (lap |V|)^2 pulled back through |.|, surface_V, the law."""
    H, B, rng = fields(43, nx=15, ny=13, holes=False)
    ph, law, th = laws(rng, "nnA")
    gl = O.Glacier(H, B, 40.0, 40.0, ph)
    Ht, tht = T(H, grad=True), T(th, grad=True)
    vx, vy = t_surface_V(Ht, T(B), 40.0, 40.0, ph, law, tht)
    V = torch.nn.functional.pad(torch.sqrt(vx * vx + vy * vy), (0, 1, 0, 1))
    mask = O.is_in_glacier(H, 2) & (V.detach().numpy() > 0.0)
    assert 20 < mask.sum() < mask.size
    L = (T(mask.astype(float)) * t_laplacian(V, 40.0, 40.0) ** 2).sum()
    l, dH, dth = O.vreg_backward(H, gl, law, th, 2)
    assert abs(L.item() - l) < 1e-11 * abs(l)
    gH, gth = torch.autograd.grad(L, [Ht, tht])
    assert rel(dH, gH.numpy()) < 1e-11 and rel(dth, gth.numpy()) < 1e-11


def test_dhdt_loss_cotangents_equal_autograd():
    """LossDhdt (TimeAggregatedLosses.jl:54-113): mask = H0 > 1e-2 is data of the evaluation, not differentiated."""
    rng = np.random.default_rng(44)
    H0 = 40.0 * rng.random((10, 9)); H0[:3] = 0.0
    H1 = H0 * (0.9 + 0.2 * rng.random((10, 9)))
    cfg = O.SimConfig(tstops=[2010.0, 2010.5, 2011.0])
    cfg.dhdt, cfg.dhdt_weight = (2010.0, 2011.0, -0.7), 2.5
    l, cot = O.dhdt_loss_terms([H0, None, H1], cfg.tstops, cfg)
    a, b = T(H0, grad=True), T(H1, grad=True)
    m = T((H0 > 1e-2).astype(float))
    L = 2.5 * ((m * (b - a)).sum() / m.sum() / 1.0 + 0.7) ** 2
    assert abs(L.item() - l) < 1e-13 * abs(l)
    g0, g1 = torch.autograd.grad(L, [a, b])
    assert rel(cot[0], g0.numpy()) < 1e-13 and rel(cot[2], g1.numpy()) < 1e-13


@pytest.mark.parametrize("filt", ["identity", "softplus", "Zang1980"])
def test_initial_condition_filters_equal_autograd(filt):
    """evaluate_H0 / evaluate_dH0 (InitialCondition_utils.jl:30-120)."""
    rng = np.random.default_rng(45)
    th = 3.0 * rng.standard_normal((8, 7))
    out = rng.random((8, 7)) > 0.7
    t = T(th, grad=True)
    if filt == "identity":
        h = t * 1.0
    elif filt == "softplus":
        h = torch.log(1.0 + torch.exp(t))
    else:
        h = torch.where(t < -1.0, torch.zeros_like(t), torch.where(t < 1.0, (t + 1.0) ** 2 / 4.0, t))
    h = h * T((~out).astype(float))
    assert rel(h.detach().numpy(), O.evaluate_H0(th, out, filt)) < 1e-14
    (g,) = torch.autograd.grad(h.sum(), [t])
    assert rel(O.evaluate_dH0(th, out, filt), g.numpy()) < 1e-14


@pytest.mark.parametrize("post", [O.POST_NONE, O.POST_AFFINE, O.POST_EXPMAX, O.POST_SCALE])
def test_mlp_theta_gradient_equals_autograd(post):
    """mlp_grad_theta stands in for Zygote's reverse pass of p_VJP! (Laws.jl:153-169,359-362): every activation and scaling."""
    rng = np.random.default_rng(46)
    m = O.MLP([2, 4, 5, 3, 2, 1], [O.ACT_SOFTPLUS, O.ACT_GELU, O.ACT_TANH, O.ACT_RELU, O.ACT_SIGMOID],
              prescale=[(-3.0, 2.0), (0.0, 7.0)], post_kind=post, post_lo=0.3, post_hi=4.2)
    th = m.init_theta(rng) + 0.2 * rng.standard_normal(m.n_params)
    X = np.stack([-3.0 + 5.0 * rng.random(6), 7.0 * rng.random(6)])
    G = O.mlp_grad_theta(m, th, X)
    tt = T(th, grad=True)
    y = t_mlp(m, tt, [T(X[0]), T(X[1])])
    assert rel(y.detach().numpy(), O.mlp_eval(m, th, X)) < 1e-14
    for k in range(6):
        (g,) = torch.autograd.grad(y[k], [tt], retain_graph=True)
        assert rel(G[:, k], g.numpy()) < 1e-13
