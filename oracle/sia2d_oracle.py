"""CPU oracle (numpy, fp64) for the SIA2D(+NN_theta) hot path of ODINN.jl.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``odinn.jl_amd/``) may
import, link or execute this file; only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may use it, and there only as the
checker.

PARITY STATUS: **forward-value parity unpinned.**  The reference is pure Julia
(no Julia toolchain in the build container) and its forward kernel
``Huginn.SIA2D!`` is an un-vendored registry dependency (Huginn compat 0.13.3,
Project.toml:84).  What pins this restatement instead:
  * the reference's own hand-written discrete adjoint re-executes the forward
    stencil line by line (src/inverse/SIA2D/adjoint.jl:52-97) -- that text is
    what ``sia2d_rhs`` below follows;
  * the reference's test identities, reproduced in tests/test_oracle_*.py:
    operator transposes (test/SIA2D_adjoint_utils.jl, rtol 1e-11), RHS-Jacobian
    vs finite differences (test/SIA2D_adjoint.jl, thresholds runtests.jl:89-91),
    full gradient vs finite differences (test/test_grad_loss.jl,
    runtests.jl:116-117), recovery of A(T) (test/inversion_test.jl:154-163) and
    the Halfar similarity solution (test/test_grad_loss.jl:498-663).

Index convention: arrays are logical ``[i, j]`` with ``i`` = x (Julia dim 1,
contiguous in the reference and in the C ABI) and ``j`` = y.  All arithmetic is
float64 (Sleipnir.Float, src/inverse/SIA2D/inversion_utils.jl:4).

Every function cites the reference file:line it restates (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

F = np.float64

# ----------------------------------------------------------------------------
# Staggered-grid operators.  Huginn.diff_x/diff_y/avg/avg_x/avg_y are out of
# tree; their semantics are pinned by the call sites in
# src/inverse/SIA2D/adjoint.jl:58-67,96-97 and by the transpose tests
# test/SIA2D_adjoint_utils.jl:18,30,96,108,120.
# ----------------------------------------------------------------------------


def diff_x(a):
    """A[i+1,j]-A[i,j]  (adjoint.jl:58 ``Huginn.diff_x(S)/dx``)."""
    return a[1:, :] - a[:-1, :]


def diff_y(a):
    """A[i,j+1]-A[i,j]  (adjoint.jl:59)."""
    return a[:, 1:] - a[:, :-1]


def avg(a):
    """4-point average onto the dual grid (adjoint.jl:67 ``Huginn.avg(H)``)."""
    return 0.25 * (a[:-1, :-1] + a[1:, :-1] + a[:-1, 1:] + a[1:, 1:])


def avg_x(a):
    """(A[i,j]+A[i+1,j])/2  (adjoint.jl:61,97)."""
    return 0.5 * (a[:-1, :] + a[1:, :])


def avg_y(a):
    """(A[i,j]+A[i,j+1])/2  (adjoint.jl:60,96)."""
    return 0.5 * (a[:, :-1] + a[:, 1:])


# --- transposes: src/inverse/SIA2D/inversion_utils.jl:3-66 -------------------


def diff_x_adjoint(I, dx):
    """inversion_utils.jl:3-8."""
    O = np.zeros((I.shape[0] + 1, I.shape[1]), F)
    O[1:, :] += I
    O[:-1, :] -= I
    return O / dx


def diff_y_adjoint(I, dy):
    """inversion_utils.jl:10-15."""
    O = np.zeros((I.shape[0], I.shape[1] + 1), F)
    O[:, 1:] += I
    O[:, :-1] -= I
    return O / dy


def clamp_borders_dx(dS, H, eta0, dx):
    """inversion_utils.jl:17-20."""
    return np.maximum(np.minimum(dS, eta0 * H[1:, 1:-1] / dx), -eta0 * H[:-1, 1:-1] / dx)


def clamp_borders_dx_adjoint(dC, eta0, dx, H, dS):
    """inversion_utils.jl:22-29 (strict inequalities).  Returns (d_dS, d_H)."""
    up = eta0 * H[1:, 1:-1] / dx
    lo = -eta0 * H[:-1, 1:-1] / dx
    d_dS = dC * ((dS < up) & (dS > lo))
    d_H = np.zeros_like(H)
    d_H[:-1, 1:-1] = -(eta0 * dC / dx) * (dS < lo)
    d_H[1:, 1:-1] += (eta0 * dC / dx) * (dS > up)
    return d_dS, d_H


def clamp_borders_dy(dS, H, eta0, dy):
    """inversion_utils.jl:31-34."""
    return np.maximum(np.minimum(dS, eta0 * H[1:-1, 1:] / dy), -eta0 * H[1:-1, :-1] / dy)


def clamp_borders_dy_adjoint(dC, eta0, dy, H, dS):
    """inversion_utils.jl:36-43."""
    up = eta0 * H[1:-1, 1:] / dy
    lo = -eta0 * H[1:-1, :-1] / dy
    d_dS = dC * ((dS < up) & (dS > lo))
    d_H = np.zeros_like(H)
    d_H[1:-1, :-1] = -(eta0 * dC / dy) * (dS < lo)
    d_H[1:-1, 1:] += (eta0 * dC / dy) * (dS > up)
    return d_dS, d_H


def avg_adjoint(I):
    """inversion_utils.jl:45-52."""
    O = np.zeros((I.shape[0] + 1, I.shape[1] + 1), F)
    O[:-1, :-1] += I
    O[1:, :-1] += I
    O[:-1, 1:] += I
    O[1:, 1:] += I
    return 0.25 * O


def avg_x_adjoint(I):
    """inversion_utils.jl:54-59."""
    O = np.zeros((I.shape[0] + 1, I.shape[1]), F)
    O[:-1, :] += I
    O[1:, :] += I
    return 0.5 * O


def avg_y_adjoint(I):
    """inversion_utils.jl:61-66."""
    O = np.zeros((I.shape[0], I.shape[1] + 1), F)
    O[:, :-1] += I
    O[:, 1:] += I
    return 0.5 * O


# ----------------------------------------------------------------------------
# MLP (Lux.Chain of Dense layers).  y = act(W x + b), W is (out x in); theta is
# flattened layer by layer as [vec(W) column-major, b] (ComponentArrays order;
# src/models/trainable_components/ML_utils.jl:31-36,54).
# ----------------------------------------------------------------------------

ACT_IDENTITY, ACT_SOFTPLUS, ACT_SIGMOID, ACT_GELU, ACT_TANH, ACT_RELU = 0, 1, 2, 3, 4, 5
POST_NONE, POST_AFFINE, POST_EXPMAX, POST_SCALE = 0, 1, 2, 3
_GELU_C = math.sqrt(2.0 / math.pi)


def _act(code, x):
    if code == ACT_IDENTITY:
        return x
    if code == ACT_SOFTPLUS:  # NNlib.softplus: log1p(exp(-|x|)) + relu(x)
        return np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0.0)
    if code == ACT_SIGMOID:  # NNlib.sigmoid (stable form)
        t = np.exp(-np.abs(x))
        return np.where(x >= 0, 1.0 / (1.0 + t), t / (1.0 + t))
    if code == ACT_GELU:  # NNlib.gelu (tanh form)
        return 0.5 * x * (1.0 + np.tanh(_GELU_C * (x + 0.044715 * x ** 3)))
    if code == ACT_TANH:
        return np.tanh(x)
    if code == ACT_RELU:
        return np.maximum(x, 0.0)
    raise ValueError(code)


def _dact(code, x):
    """d act / d x (pre-activation x)."""
    if code == ACT_IDENTITY:
        return np.ones_like(x)
    if code == ACT_SOFTPLUS:
        return _act(ACT_SIGMOID, x)
    if code == ACT_SIGMOID:
        s = _act(ACT_SIGMOID, x)
        return s * (1.0 - s)
    if code == ACT_GELU:
        u = _GELU_C * (x + 0.044715 * x ** 3)
        th = np.tanh(u)
        du = _GELU_C * (1.0 + 3 * 0.044715 * x ** 2)
        return 0.5 * (1.0 + th) + 0.5 * x * (1.0 - th * th) * du
    if code == ACT_TANH:
        return 1.0 - np.tanh(x) ** 2
    if code == ACT_RELU:
        return (x > 0).astype(F)
    raise ValueError(code)


@dataclass
class MLP:
    """Descriptor of a Lux.Chain(Dense...) with ODINN's pre/post scaling.

    prescale: per-input (lo, hi) -> (x-lo)/(hi-lo)-0.5
      (src/models/target/target_utils.jl:58-64,131-141) or None.
    postscale: POST_AFFINE -> lo+(hi-lo)*y (target_utils.jl:109-113, Laws.jl:351);
      POST_EXPMAX -> hi*exp((y-1)/y) (target_utils.jl:86-93); POST_SCALE -> hi*y.
    """

    widths: Sequence[int]  # [n_in, h1, ..., n_out=1]
    acts: Sequence[int]  # one per Dense layer
    prescale: Optional[Sequence[Tuple[float, float]]] = None
    post_kind: int = POST_NONE
    post_lo: float = 0.0
    post_hi: float = 1.0

    @property
    def n_in(self):
        return self.widths[0]

    @property
    def n_params(self):
        return sum(self.widths[l + 1] * (self.widths[l] + 1) for l in range(len(self.acts)))

    def unpack(self, theta):
        out, o = [], 0
        for l in range(len(self.acts)):
            nin, nout = self.widths[l], self.widths[l + 1]
            W = np.asarray(theta[o : o + nin * nout], F).reshape((nout, nin), order="F")
            o += nin * nout
            b = np.asarray(theta[o : o + nout], F)
            o += nout
            out.append((W, b))
        assert o == len(theta)
        return out

    def init_theta(self, rng):
        """Glorot-uniform weights / zero bias (Lux default init), own RNG stream."""
        th = []
        for l in range(len(self.acts)):
            nin, nout = self.widths[l], self.widths[l + 1]
            lim = math.sqrt(6.0 / (nin + nout))
            th.append(rng.uniform(-lim, lim, size=nin * nout))
            th.append(np.zeros(nout))
        return np.concatenate(th).astype(F)


def default_nn(n_input=1, light=False, **kw):
    """build_default_NN (src/models/trainable_components/ML_utils.jl:23-39)."""
    if light:
        return MLP([n_input, 3, 1], [ACT_SOFTPLUS, ACT_SIGMOID], **kw)
    return MLP([n_input, 3, 10, 3, 1], [ACT_SOFTPLUS, ACT_SOFTPLUS, ACT_SOFTPLUS, ACT_SIGMOID], **kw)


def _prescale(mlp, X):
    if mlp.prescale is None:
        return X
    lo = np.array([b[0] for b in mlp.prescale], F).reshape((-1,) + (1,) * (X.ndim - 1))
    hi = np.array([b[1] for b in mlp.prescale], F).reshape((-1,) + (1,) * (X.ndim - 1))
    return (X - lo) / (hi - lo) - 0.5


def _postscale(mlp, y):
    if mlp.post_kind == POST_NONE:
        return y
    if mlp.post_kind == POST_AFFINE:
        return mlp.post_lo + (mlp.post_hi - mlp.post_lo) * y
    if mlp.post_kind == POST_EXPMAX:
        return mlp.post_hi * np.exp((y - 1.0) / y)
    if mlp.post_kind == POST_SCALE:
        return mlp.post_hi * y
    raise ValueError(mlp.post_kind)


def _dpostscale(mlp, y):
    if mlp.post_kind == POST_NONE:
        return np.ones_like(y)
    if mlp.post_kind == POST_AFFINE:
        return np.full_like(y, mlp.post_hi - mlp.post_lo)
    if mlp.post_kind == POST_EXPMAX:
        return mlp.post_hi * np.exp((y - 1.0) / y) / (y * y)
    if mlp.post_kind == POST_SCALE:
        return np.full_like(y, mlp.post_hi)
    raise ValueError(mlp.post_kind)


def mlp_eval(mlp: MLP, theta, X):
    """_pred_NN (src/laws/Laws.jl:34-36), vectorised: X has shape (n_in, ...)."""
    X = np.asarray(X, F)
    h = _prescale(mlp, X)
    for (W, b), a in zip(mlp.unpack(theta), mlp.acts):
        z = np.tensordot(W, h, axes=(1, 0)) + b.reshape((-1,) + (1,) * (h.ndim - 1))
        h = _act(a, z)
    return _postscale(mlp, h[0])


def mlp_grad_theta(mlp: MLP, theta, X):
    """d out / d theta at each input point: shape (P, ...).  Stands in for the
    Zygote/Mooncake reverse pass of ``p_VJP!`` / ``dlaw/dtheta`` (Laws.jl:359-362,
    src/laws/auto_VJP.jl:114-122) -- exact analytic backprop."""
    X = np.asarray(X, F)
    h = _prescale(mlp, X)
    layers = mlp.unpack(theta)
    hs, zs = [h], []
    for (W, b), a in zip(layers, mlp.acts):
        z = np.tensordot(W, h, axes=(1, 0)) + b.reshape((-1,) + (1,) * (h.ndim - 1))
        zs.append(z)
        h = _act(a, z)
        hs.append(h)
    g = _dpostscale(mlp, h[0])[None]  # d out / d h_L
    grads = []
    for l in reversed(range(len(layers))):
        W, b = layers[l]
        dz = g * _dact(mlp.acts[l], zs[l])  # (nout, ...)
        dW = dz[:, None] * hs[l][None, :]  # (nout, nin, ...)
        nout, nin = W.shape
        dWf = np.transpose(dW, (1, 0) + tuple(range(2, dW.ndim))).reshape((nin * nout,) + dW.shape[2:])
        grads.append((dWf, dz))
        g = np.tensordot(W.T, dz, axes=(1, 0))
    out = []
    for dWf, db in reversed(grads):
        out.append(dWf)
        out.append(db)
    return np.concatenate(out, axis=0)


# ----------------------------------------------------------------------------
# Physical parameters / glacier / law configuration
# ----------------------------------------------------------------------------


@dataclass
class Phys:
    """Sleipnir.PhysicalParameters fields used on the path (test/params_construction.jl:24-34)."""

    rho: float = 900.0
    g: float = 9.81
    eta0: float = 1.0
    n: float = 3.0
    p: float = 3.0  # sliding exponents (OOT defaults unknown; inputs here)
    q: float = 0.0
    C: float = 0.0
    minA: float = 8e-21
    maxA: float = 8e-17


LAW_CONST_A, LAW_NN_A_SCALAR, LAW_NN_A_GRIDDED, LAW_NN_Y, LAW_NN_U = 0, 1, 2, 3, 4


@dataclass
class Law:
    """Which quantity theta drives.

    CONST_A      : A given (scalar or (nx-1,ny-1) dual field); no theta.
    NN_A_SCALAR  : A = minA+(maxA-minA)*MLP(T), scalar T (Laws.jl:348-358), target :A
    NN_A_GRIDDED : same on a gridded T (dual grid here), target :A
    NN_Y         : Y = maxNN*exp((y-1)/y), y = MLP(norm(T), norm(Hbar)) (Laws.jl:258-265), target :D_hybrid
    NN_U         : U = post(MLP(pre(Hbar, gradS))) (Laws.jl:114-123), D = Hbar*U, target :D
    """

    kind: int = LAW_CONST_A
    mlp: Optional[MLP] = None
    theta: Optional[np.ndarray] = None
    A: object = 2.21e-18  # CONST_A value
    T: object = -5.0  # temperature input (scalar or dual-grid field)
    n_H: Optional[float] = None  # D_hybrid exponents (target_D_hybrid.jl:180-185)
    n_gradS: Optional[float] = None
    # spatial evaluation of d law / d theta (SIA2D_D_hybrid_target.interpolation, target_D_hybrid.jl:12-15;
    # SIA2D_D_target, target_D_pure.jl:34-39): "none" = exact per node, "linear" = gradients on 2 n_interp_half nodes of
    # Hbar, linearly interpolated.  None = the reference's default for the law: :Linear with n_interp_half = 75 for the Y
    # law (:D_hybrid), :None for every other law.
    interpolation: Optional[str] = None
    n_interp_half: Optional[int] = None
    # parameters.simulation.f_surface_velocity_factor (target :D: Velocity^ = U / f, target_D_pure.jl:206-255); kept with the
    # law here because only the U law reads it
    fV: float = 1.0

    def interp(self):
        kind = self.interpolation if self.interpolation is not None else ("linear" if self.kind == LAW_NN_Y else "none")
        n = self.n_interp_half if self.n_interp_half is not None else (75 if self.kind == LAW_NN_Y else 100)
        return kind, n


@dataclass
class Glacier:
    H0: np.ndarray
    B: np.ndarray
    dx: float
    dy: float
    phys: Phys = field(default_factory=Phys)

    @property
    def shape(self):
        return self.B.shape


def gamma_no_A(ph: Phys):
    """Gamma(...; include_A=false) = 2 (rho g)^n/(n+2)  (target_utils.jl:3-12)."""
    return 2.0 * (ph.rho * ph.g) ** ph.n / (ph.n + 2.0)


def sliding_S(ph: Phys):
    """S = C (rho g)^(p-q)  (target_utils.jl:14-18)."""
    return ph.C * (ph.rho * ph.g) ** (ph.p - ph.q)


def _pow(x, e):
    """Julia ``x .^ e`` with float exponent (0^0 == 1)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.power(x, e)


def law_value(law: Law, ph: Phys, Hbar, gradS, theta=None):
    """Evaluate the law on the dual grid (apply_all_non_callback_laws!, adjoint.jl:75-76;
    LawA Laws.jl:348-358, LawY :258-265, LawU :114-123)."""
    th = law.theta if theta is None else theta
    if law.kind == LAW_CONST_A:
        return law.A
    if law.kind == LAW_NN_A_SCALAR:
        return mlp_eval(law.mlp, th, np.array([[float(law.T)]], F))[0]
    if law.kind == LAW_NN_A_GRIDDED:
        return mlp_eval(law.mlp, th, np.asarray(law.T, F)[None])
    if law.kind == LAW_NN_Y:
        T = np.full_like(Hbar, law.T) if np.ndim(law.T) == 0 else law.T
        return mlp_eval(law.mlp, th, np.stack([T, Hbar]))
    if law.kind == LAW_NN_U:
        return mlp_eval(law.mlp, th, np.stack([Hbar, gradS]))
    raise ValueError(law.kind)


def _hyb_exps(law: Law, ph: Phys):
    nH = ph.n if law.n_H is None else law.n_H
    nS = ph.n if law.n_gradS is None else law.n_gradS
    return nH, nS


def diffusivity(law: Law, ph: Phys, Hbar, gradS, theta=None):
    """Diffusivity for targets :A (target_A.jl:16-30), :D_hybrid
    (target_D_hybrid.jl:168-208), :D (target_D_pure.jl:78-96)."""
    val = law_value(law, ph, Hbar, gradS, theta)
    if law.kind == LAW_NN_U:
        return Hbar * val
    if law.kind == LAW_NN_Y:
        nH, nS = _hyb_exps(law, ph)
    else:
        nH, nS = ph.n, ph.n
    G = gamma_no_A(ph)
    Sc = sliding_S(ph)
    D = val * G * _pow(Hbar, nH + 2.0) * _pow(gradS, nS - 1.0)
    if Sc != 0.0:
        D = D + Sc * _pow(Hbar, ph.p - ph.q + 1.0) * _pow(gradS, ph.p - 1.0)
    return D


def d_diffusivity_dH(law: Law, ph: Phys, Hbar, gradS, theta=None):
    """alpha = dD/dHbar.  :A target_A.jl:32-46; :D_hybrid target_D_hybrid.jl:34-74
    (NN part by forward FD, dH=1e-4, :58-71); :D central FD dH=1e-4 masked by
    Hbar>0 (target_D_pure.jl:105-120)."""
    G = gamma_no_A(ph)
    Sc = sliding_S(ph)
    if law.kind == LAW_NN_U:
        dH = 1e-4
        Dp = law_value(law, ph, Hbar + dH, gradS, theta) * (Hbar + dH)
        Dm = law_value(law, ph, Hbar - dH, gradS, theta) * (Hbar - dH)
        return (Hbar > 0.0) * ((Dp - Dm) / (2.0 * dH))
    val = law_value(law, ph, Hbar, gradS, theta)
    if law.kind == LAW_NN_Y:
        nH, nS = _hyb_exps(law, ph)
    else:
        nH, nS = ph.n, ph.n
    out = val * G * (nH + 2.0) * _pow(Hbar, nH + 1.0) * _pow(gradS, nS - 1.0)
    if Sc != 0.0:
        out = out + (ph.p - ph.q + 1.0) * Sc * _pow(Hbar, ph.p - ph.q) * _pow(gradS, ph.p - 1.0)
    if law.kind == LAW_NN_Y:
        dH = 1e-4
        Yp = law_value(law, ph, Hbar + dH, gradS, theta)
        geo = G * _pow(Hbar, nH + 2.0) * _pow(gradS, nS - 1.0)
        slide = Sc * _pow(Hbar, ph.p - ph.q + 1.0) * _pow(gradS, ph.p - 1.0) if Sc != 0.0 else 0.0
        a = slide + Yp * geo
        b = slide + val * geo
        out = out + (a - b) / dH
    return out


def d_diffusivity_dgradS(law: Law, ph: Phys, Hbar, gradS, theta=None):
    """beta = (dD/d|gradS|)/|gradS| as the reference writes it (note gradS^(n-3)).
    :A target_A.jl:48-62; :D_hybrid target_D_hybrid.jl:76-96; :D central FD
    d=1e-6 (target_D_pure.jl:123-137)."""
    if law.kind == LAW_NN_U:
        d = 1e-6
        Dp = law_value(law, ph, Hbar, gradS + d, theta) * Hbar
        Dm = law_value(law, ph, Hbar, gradS - d, theta) * Hbar
        return (Dp - Dm) / (2.0 * d)
    G = gamma_no_A(ph)
    Sc = sliding_S(ph)
    val = law_value(law, ph, Hbar, gradS, theta)
    if law.kind == LAW_NN_Y:
        nH, nS = _hyb_exps(law, ph)
    else:
        nH, nS = ph.n, ph.n
    out = val * G * (nS - 1.0) * _pow(Hbar, nH + 2.0) * _pow(gradS, nS - 3.0)
    if Sc != 0.0:
        out = out + Sc * (ph.p - 1.0) * _pow(Hbar, ph.p - ph.q + 1.0) * _pow(gradS, ph.p - 3.0)
    return out


# ----------------------------------------------------------------------------
# Forward RHS  (Huginn.SIA2D!, restated from adjoint.jl:52-97)
# ----------------------------------------------------------------------------


def _forward_intermediates(H, B, dx, dy, ph: Phys):
    H = np.where(H > 0.0, H, 0.0)  # adjoint.jl:52
    S = B + H  # :54
    dSdx = diff_x(S) / dx  # :58
    dSdy = diff_y(S) / dy  # :59
    gSx = avg_y(dSdx)  # :60
    gSy = avg_x(dSdy)  # :61
    gS = (gSx ** 2 + gSy ** 2) ** 0.5  # :64
    Hbar = avg(H)  # :67
    ex = diff_x(S[:, 1:-1]) / dx  # :87
    ey = diff_y(S[1:-1, :]) / dy  # :88
    exc = clamp_borders_dx(ex, H, ph.eta0, dx)  # :93
    eyc = clamp_borders_dy(ey, H, ph.eta0, dy)  # :94
    return H, S, gSx, gSy, gS, Hbar, ex, ey, exc, eyc


def sia2d_rhs(H, B, dx, dy, ph: Phys, law: Law, theta=None):
    """dH/dt = div(D grad S) on the interior, 0 on the boundary ring."""
    Hc, S, gSx, gSy, gS, Hbar, ex, ey, exc, eyc = _forward_intermediates(H, B, dx, dy, ph)
    D = diffusivity(law, ph, Hbar, gS, theta)  # adjoint.jl:79-84
    Dx = avg_y(D)  # :96
    Dy = avg_x(D)  # :97
    Fx = -Dx * exc
    Fy = -Dy * eyc
    dH = np.zeros_like(Hc)
    dH[1:-1, 1:-1] = -(diff_x(Fx) / dx + diff_y(Fy) / dy)
    return dH


def _D_adjoint(lam, dx, dy, exc, eyc):
    """adjoint.jl:99-104 / :235-240."""
    li = lam[1:-1, 1:-1]
    Fxa = diff_x_adjoint(-li, dx)
    Fya = diff_y_adjoint(-li, dy)
    Da = avg_y_adjoint(-Fxa * exc) + avg_x_adjoint(-Fya * eyc)
    return Fxa, Fya, Da


def vjp_H(lam, H, B, dx, dy, ph: Phys, law: Law, theta=None):
    """VJP_lambda_dSIA/dH_discrete (adjoint.jl:31-151)."""
    Hc, S, gSx, gSy, gS, Hbar, ex, ey, exc, eyc = _forward_intermediates(H, B, dx, dy, ph)
    D = diffusivity(law, ph, Hbar, gS, theta)
    Dx, Dy = avg_y(D), avg_x(D)
    Fxa, Fya, Da = _D_adjoint(lam, dx, dy, exc, eyc)
    alpha = d_diffusivity_dH(law, ph, Hbar, gS, theta)  # :109-114
    beta = d_diffusivity_dgradS(law, ph, Hbar, gS, theta)  # :116-121
    bx = beta * gSx
    by = beta * gSy
    dDdH = (
        avg_adjoint(alpha * Da)
        + diff_x_adjoint(avg_y_adjoint(bx * Da), dx)
        + diff_y_adjoint(avg_x_adjoint(by * Da), dy)
    )  # :125-127
    dCx = -Fxa * Dx  # :130
    dCy = -Fya * Dy  # :131
    d_ex, dHx = clamp_borders_dx_adjoint(dCx, ph.eta0, dx, Hc, ex)  # :136
    d_ey, dHy = clamp_borders_dy_adjoint(dCy, ph.eta0, dy, Hc, ey)  # :137
    gx = np.zeros_like(S)
    gx[:, 1:-1] = diff_x_adjoint(d_ex, dx)  # :138-139
    gy = np.zeros_like(S)
    gy[1:-1, :] = diff_y_adjoint(d_ey, dy)  # :141-142
    dlam = dDdH + gx + dHx + gy + dHy  # :140-147
    return dlam * (Hc > 0.0)  # :148


def dD_dlaw(law: Law, ph: Phys, Hbar, gradS):
    """The spatial factor of dD/dtheta: Gamma0 Hbar^(n+2) gradS^(n-1) for :A
    (target_A.jl:71-72) / :D_hybrid (target_D_hybrid.jl:117-118); Hbar*[Hbar>0]
    for :D (target_D_pure.jl:142,159)."""
    if law.kind == LAW_NN_U:
        return Hbar * (Hbar > 0.0)
    nH, nS = _hyb_exps(law, ph) if law.kind == LAW_NN_Y else (ph.n, ph.n)
    return gamma_no_A(ph) * _pow(Hbar, nH + 2.0) * _pow(gradS, nS - 1.0)


def law_grad_theta(law: Law, ph: Phys, Hbar, gradS, theta=None):
    """d law / d theta on the dual grid, shape (P, nx-1, ny-1) (scalar law: (P,)).
    Exact per-node evaluation == the reference's ``interpolation = :None`` branch
    (target_D_hybrid.jl:121-131, target_D_pure.jl:163-176)."""
    th = law.theta if theta is None else theta
    if law.kind == LAW_NN_A_SCALAR:
        return mlp_grad_theta(law.mlp, th, np.array([[float(law.T)]], F))[:, 0]
    if law.kind == LAW_NN_A_GRIDDED:
        return mlp_grad_theta(law.mlp, th, np.asarray(law.T, F)[None])
    if law.kind == LAW_NN_Y:
        T = np.full_like(Hbar, law.T) if np.ndim(law.T) == 0 else law.T
        return mlp_grad_theta(law.mlp, th, np.stack([T, Hbar]))
    if law.kind == LAW_NN_U:
        return mlp_grad_theta(law.mlp, th, np.stack([Hbar, gradS]))
    raise ValueError(law.kind)


def create_interpolation(A, n_interp_half, dilation_factor=1.0, minA_unif=None, minA_quantile=None, maxA_unif=None,
                         maxA_quantile=None):
    """create_interpolation (src/models/target/target_utils.jl:245-293): the sorted, unique union of n_interp_half
    uniformly spaced values on [minA_unif, maxA_unif] (default [0, dilation * max A]) and n_interp_half quantiles
    (probabilities LinRange(0, 1, n + 2)[2:end-1], Statistics.quantile's default definition = type 7) of the entries of A
    strictly inside (minA_quantile, maxA_quantile) (default (0, max A)).
    The reference tops the vector up to exactly 2 n_interp_half knots with midpoints of RANDOMLY chosen intervals when
    the union has duplicates ("in theory, this should never happen", :278-288); that step is not reproducible and is
    omitted: the vector then simply has fewer knots."""
    A = np.asarray(A, F).ravel()
    amax = A.max()
    lo_u = 0.0 if minA_unif is None else minA_unif
    lo_q = 0.0 if minA_quantile is None else minA_quantile
    hi_u = dilation_factor * amax if maxA_unif is None else maxA_unif
    hi_q = amax if maxA_quantile is None else maxA_quantile
    if not (lo_u < hi_u and lo_q < hi_q):
        raise ValueError("There are not enough different values of A to create a proper interpolation.")  # :262
    n = int(n_interp_half)
    t = np.arange(n) / (n - 1.0)
    unif = (1.0 - t) * lo_u + t * hi_u  # LinRange: lerp with t = j / (n - 1)
    probs = (np.arange(n + 2) / (n + 1.0))[1:-1]
    inside = np.sort(A[(lo_q < A) & (A < hi_q)])
    if inside.size == 0:
        return np.unique(unif)
    h = (inside.size - 1) * probs  # 0-based position; Statistics._quantile: aleph = (m - 1) p + 1, j = trunc(aleph), gamma
    j = np.clip(np.floor(h).astype(int), 0, max(inside.size - 2, 0))
    gam = np.clip(h - j, 0.0, 1.0)
    a = inside[j]
    b = inside[np.minimum(j + 1, inside.size - 1)]
    quant = a + gam * (b - a)
    return np.unique(np.concatenate([unif, quant]))  # sorted + unique


def interp_linear_weights(nodes, x):
    """Gridded(Linear()) interpolation (Interpolations.jl) of knot values at x in [nodes[0], nodes[-1]]: returns
    (k, w) with value = (1 - w) v[k] + w v[k + 1]."""
    x = np.asarray(x, F)
    k = np.clip(np.searchsorted(nodes, x, side="right") - 1, 0, len(nodes) - 2)
    w = (x - nodes[k]) / (nodes[k + 1] - nodes[k])
    return k, w


def law_grad_theta_linear(law: Law, ph: Phys, Hbar, gradS, theta=None, n_interp_half=75):
    """The `interpolation == :Linear` branch of dDiffusivity/dtheta for the Y law (target_D_hybrid.jl:136-160; the same
    text serves dVelocity/dtheta, :321-345): d law / d theta evaluated exactly on the knots of create_interpolation(Hbar)
    and interpolated linearly in Hbar at every dual node.  Shape (P, nx-1, ny-1)."""
    if law.kind != LAW_NN_Y:
        raise ValueError("linear interpolation of the law gradient is restated for the Y law (:D_hybrid) only")
    th = law.theta if theta is None else theta
    nodes = create_interpolation(Hbar, n_interp_half)
    T = np.full_like(nodes, float(law.T))
    G = mlp_grad_theta(law.mlp, th, np.stack([T, nodes]))  # (P, M): exact gradients at the knots
    if nodes.size == 1:
        return np.broadcast_to(G[:, :1, None], (G.shape[0],) + Hbar.shape).copy()
    k, w = interp_linear_weights(nodes, Hbar)
    return (1.0 - w)[None] * G[:, k] + w[None] * G[:, k + 1]


U_INTERP_NODE_MAX = 100.0  # LinRange(0.0, 100, n_nodes) (Laws.jl:131)


def law_grad_theta_bilinear(law: Law, ph: Phys, Hbar, gradS, theta=None, n_interp_half=100):
    """The `interpolation == :Linear` branch of dU/dtheta for the U law (target_D_pure.jl:179-193): d law / d theta
    evaluated exactly on a fixed n_nodes x n_nodes grid, n_nodes = 2 n_interp_half (p_VJP!, Laws.jl:153-169), and
    interpolated bilinearly (Interpolations.Gridded(Linear())) at (Hbar, |grad S|) of every dual node.
    Both node axes are LinRange(0, 100, n_nodes): the law's cache is constructed with the Hbar nodes in the place of the
    slope nodes as well (MatrixCacheInterp(zeros(..), H_nodes, H_nodes, grad_itp), Laws.jl:137-142), so the slope axis
    the reference interpolates on is [0, 100] too and every physical slope lies in its first interval.  A gridded
    interpolant does not extrapolate: a node with Hbar > 100 raises (BoundsError in the reference).
    Shape (P, nx-1, ny-1)."""
    if law.kind != LAW_NN_U:
        raise ValueError("bilinear interpolation of the law gradient is the U law's (:D) branch")
    th = law.theta if theta is None else theta
    K = 2 * int(n_interp_half)
    t = np.arange(K) / (K - 1.0)
    nodes = (1.0 - t) * 0.0 + t * U_INTERP_NODE_MAX
    if Hbar.max() > nodes[-1] or gradS.max() > nodes[-1] or Hbar.min() < 0.0 or gradS.min() < 0.0:
        raise IndexError("BoundsError: the gradient interpolant of the U law is defined on [0, 100] x [0, 100]")
    hh, ss = np.meshgrid(nodes, nodes, indexing="ij")
    G = mlp_grad_theta(law.mlp, th, np.stack([hh, ss]))  # (P, K, K): exact gradients on the node grid
    kh, wh = interp_linear_weights(nodes, Hbar)
    ks, ws = interp_linear_weights(nodes, gradS)
    return ((1.0 - wh) * (1.0 - ws))[None] * G[:, kh, ks] + (wh * (1.0 - ws))[None] * G[:, kh + 1, ks] \
        + ((1.0 - wh) * ws)[None] * G[:, kh, ks + 1] + (wh * ws)[None] * G[:, kh + 1, ks + 1]


def vjp_theta(lam, H, B, dx, dy, ph: Phys, law: Law, theta=None):
    """VJP_lambda_dSIA/dtheta_discrete (adjoint.jl:178-255):
    dtheta_k = sum_ij dD/dtheta_k[i,j] * D_adjoint[i,j]."""
    Hc, S, gSx, gSy, gS, Hbar, ex, ey, exc, eyc = _forward_intermediates(H, B, dx, dy, ph)
    _, _, Da = _D_adjoint(lam, dx, dy, exc, eyc)
    spatial = dD_dlaw(law, ph, Hbar, gS) * Da
    if law.kind == LAW_CONST_A:
        return np.array([np.sum(spatial)])  # d/dA (one "parameter": A itself)
    kind, nhalf = law.interp()
    if kind == "linear" and law.kind == LAW_NN_Y and Hbar.max() > 0.0:
        g = law_grad_theta_linear(law, ph, Hbar, gS, theta, nhalf)  # the reference's default for :D_hybrid
    elif kind == "linear" and law.kind == LAW_NN_U:
        g = law_grad_theta_bilinear(law, ph, Hbar, gS, theta, nhalf)  # SIA2D_D_target(interpolation = :Linear)
    else:
        g = law_grad_theta(law, ph, Hbar, gS, theta)
    if law.kind == LAW_NN_A_SCALAR:
        return g.reshape(-1) * np.sum(spatial)  # cartesian_tensor (target_utils.jl:156-161)
    return np.tensordot(g, spatial, axes=([1, 2], [0, 1]))


def vjp_H_continuous(lam, H, B, dx, dy, ph: Phys, law: Law, theta=None):
    """VJP_lambda_dSIA/dH_continuous (adjoint.jl:442-553): the continuous-form adjoint
    div(D grad lam) - dD/dH <grad S, grad lam> + div(dD/dgradH <grad S, grad lam>) on the interior,
    with unclamped slopes and no H > 0 mask; 0 on the boundary ring."""
    Hc = np.maximum(H, 0.0)
    S = B + Hc
    dSdx = diff_x(S) / dx
    dSdy = diff_y(S) / dy
    gSx = avg_y(dSdx)
    gSy = avg_x(dSdy)
    gS = np.sqrt(gSx ** 2 + gSy ** 2)
    Hbar = avg(Hc)
    D = diffusivity(law, ph, Hbar, gS, theta)
    dDdH = avg(d_diffusivity_dH(law, ph, Hbar, gS, theta))  # :499-505
    beta = d_diffusivity_dgradS(law, ph, Hbar, gS, theta)
    dDdgx = beta * gSx  # :515-516
    dDdgy = beta * gSy
    dlx = diff_x(lam[:, 1:-1]) / dx  # :521-522
    dly = diff_y(lam[1:-1, :]) / dy
    Fx = -avg_y(D) * dlx
    Fy = -avg_x(D) * dly
    divDgl = -(diff_x(Fx) / dx + diff_y(Fy) / dy)  # :527-531
    glgS = avg_y(dSdx * diff_x(lam) / dx) + avg_x(dSdy * diff_y(lam) / dy)  # :534-538
    t2 = dDdH * avg(glgS)  # :540
    px = glgS * dDdgx
    py = glgS * dDdgy
    t3 = avg_y(diff_x(px) / dx) + avg_x(diff_y(py) / dy)  # :543-548
    out = np.zeros_like(lam)
    out[1:-1, 1:-1] = divDgl - t2 + t3  # :551-552
    return out


def vjp_theta_continuous(lam, H, B, dx, dy, ph: Phys, law: Law, theta=None):
    """VJP_lambda_dSIA/dtheta_continuous (adjoint.jl:583-662): sum_ij lam[i,j] * div(avg(dD/dtheta_k) *
    clamped grad S)[i,j] -- the forward form of the same bilinear expression as the discrete VJP
    (the two agree to rounding; only the order of summation differs)."""
    Hc, S, gSx, gSy, gS, Hbar, ex, ey, exc, eyc = _forward_intermediates(H, B, dx, dy, ph)
    spat = dD_dlaw(law, ph, Hbar, gS)

    def contract(w):  # w = dD/dtheta_k on the dual grid
        Fx = avg_y(w) * exc
        Fy = avg_x(w) * eyc
        return float(np.sum((diff_x(Fx) / dx + diff_y(Fy) / dy) * lam[1:-1, 1:-1]))

    if law.kind == LAW_CONST_A:
        return np.array([contract(spat)])
    # dDiffusivity/dtheta(target; ...) (:634-639) is the same call as in the discrete VJP: it honours target.interpolation
    kind, nhalf = law.interp()
    if kind == "linear" and law.kind == LAW_NN_Y and Hbar.max() > 0.0:
        g = law_grad_theta_linear(law, ph, Hbar, gS, theta, nhalf)
    elif kind == "linear" and law.kind == LAW_NN_U:
        g = law_grad_theta_bilinear(law, ph, Hbar, gS, theta, nhalf)
    else:
        g = law_grad_theta(law, ph, Hbar, gS, theta)
    if law.kind == LAW_NN_A_SCALAR:
        return g.reshape(-1) * contract(spat)
    return np.array([contract(spat * g[k]) for k in range(g.shape[0])])


# ----------------------------------------------------------------------------
# Mass balance source (callback inversion_utils.jl:498-517; mask/clip logic
# mirrored at src/inverse/SIA2D/VJPs.jl:129-139).  The climate model (Muninn
# TImodel1) is out of tree: the MB *increment over one step_MB* is an input
# here, optionally with an elevation feedback dMB/dS (own synthetic stand-in
# for the PDD lapse-rate term of VJPs.jl:119-123,147).
# ----------------------------------------------------------------------------


@dataclass
class MassBalance:
    mb0: np.ndarray  # MB increment per step_MB at the reference surface S_ref [m]
    dmb_dS: float = 0.0  # elevation feedback [m per m per step_MB]
    S_ref: Optional[np.ndarray] = None
    mb_max: float = np.inf  # accumulation cap (the "snow" part has zero H-derivative)


def mb_raw(mb: MassBalance, H, B):
    if mb.dmb_dS == 0.0:
        return mb.mb0.copy(), np.zeros_like(H)
    raw = mb.mb0 + mb.dmb_dS * ((B + H) - mb.S_ref)
    sat = raw >= mb.mb_max
    return np.where(sat, mb.mb_max, raw), np.where(sat, 0.0, mb.dmb_dS)


def mb_apply(mb: MassBalance, H, B):
    """Returns (H_new, MB_applied) -- VJPs.jl:129-139 + apply_MB_mask!."""
    MB, _ = mb_raw(mb, H, B)
    mask = ((H > 0.0) & (MB < 0.0)) | ((H > 10.0) & (MB >= 0.0))  # :129
    MB = np.where(mask, MB, 0.0)  # :131
    dis = mask & ((H + MB) < 0.0)  # :133-137
    MB = np.where(dis, -H, MB)  # :139
    return H + MB, MB


def vjp_mb(mb: MassBalance, lam, H_pre, B):
    """VJP_lambda_dMB/dH (DiscreteVJP, VJPs.jl:107-151): diagonal Jacobian of the
    applied MB w.r.t. the pre-MB thickness."""
    MB, dMB = mb_raw(mb, H_pre, B)
    mask = ((H_pre > 0.0) & (MB < 0.0)) | ((H_pre > 10.0) & (MB >= 0.0))
    MBm = np.where(mask, MB, 0.0)
    dis = mask & ((H_pre + MBm) < 0.0)
    out = np.where(mask, dMB * lam, 0.0)  # :147
    out = np.where(dis, -lam, out)  # :148
    return out


# ----------------------------------------------------------------------------
# Loss: LossH(L2Sum(distance))  (src/losses/Losses.jl:133-152,250-291)
# ----------------------------------------------------------------------------


def is_in_glacier(Href, distance):
    """Sleipnir.is_in_glacier is out of tree (Losses.jl:122,266).  Own definition
    (flagged in DESIGN.md): a cell is in the glacier iff Href>0 there and at
    every cell within Chebyshev distance ``distance``."""
    m = Href > 0.0
    if distance <= 0:
        return m
    nx, ny = m.shape
    pad = np.zeros((nx + 2 * distance, ny + 2 * distance), bool)
    pad[distance : distance + nx, distance : distance + ny] = m
    out = np.ones_like(m)
    for a in range(2 * distance + 1):
        for b in range(2 * distance + 1):
            out &= pad[a : a + nx, b : b + ny]
    return out


def l2sum_loss(a, b, mask, normalization):
    """Losses.jl:133-141."""
    return np.sum(((a - b)[mask]) ** 2) / normalization


def l2sum_backward(a, b, mask, normalization):
    """Losses.jl:142-152."""
    d = np.zeros_like(a)
    d[mask] = a[mask] - b[mask]
    return 2.0 * d / normalization


def logsum_loss(a, b, mask, normalization, eps):
    """LogSum (Losses.jl:207-217): sum log^2((a + eps) / (b + eps)) / normalization on the mask; a, b >= 0 asserted."""
    assert a.min() >= 0.0 and b.min() >= 0.0
    return np.sum(np.log((a[mask] + eps) / (b[mask] + eps)) ** 2) / normalization


def logsum_backward(a, b, mask, normalization, eps):
    """Losses.jl:218-229."""
    d = np.zeros_like(a)
    d[mask] = np.log((a[mask] + eps) / (b[mask] + eps)) / (a[mask] + eps)
    return 2.0 * d / normalization


def hloss(a, b, mask, normalization, eps=None):
    """The simple loss inside LossH: L2Sum (eps None) or LogSum(eps)."""
    return l2sum_loss(a, b, mask, normalization) if eps is None else logsum_loss(a, b, mask, normalization, eps)


def hloss_backward(a, b, mask, normalization, eps=None):
    return l2sum_backward(a, b, mask, normalization) if eps is None else logsum_backward(a, b, mask, normalization, eps)


# ----------------------------------------------------------------------------
# Time integration: RDPK3Sp35 (3S*+ low-storage, 5 stages, order 3(2)),
# Ranocha, Dalcin, Parsani, Ketcheson (2022) "Optimized Runge-Kutta methods with
# automatic step size control for compressible CFD", with the PID controller
# (0.64, -0.31, 0.04) recommended there.  This is the solver the reference's
# gradient tests select (test/test_grad_loss.jl:143) through OrdinaryDiffEq
# (compat "6", Project.toml:105; third-party, not in /root/reference).  The
# coefficient set below satisfies the row-sum and all four 3rd-order conditions
# to 1e-37 (tests/test_oracle_integrator.py).
# ----------------------------------------------------------------------------

RDPK_G1 = (0.0, 2.587771979725733308135192812685323706e-01, -1.324380360140723382965420909764953437e-01,
           5.056033948190826045833606441415585735e-02, 5.670532000739313812633197158607642990e-01)
RDPK_G2 = (1.0, 5.528354909301389892439698870483746541e-01, 6.731871608203061824849561782794643600e-01,
           2.803103963297672407841316576323901761e-01, 5.521525447020610386070346724931300367e-01)
RDPK_G3 = (0.0, 0.0, 0.0, 2.752563273304676380891217287572780582e-01, -8.950526174674033822276061734289327568e-01)
RDPK_DELTA = (1.0, 3.407655879334525365094815965895763636e-01, 3.414382655003386206551709871126405331e-01,
              7.229275366787987419692007421895451953e-01, 0.0)
RDPK_BETA = (2.300298624518076223899418286314123354e-01, 3.021434166948288809034402119555380003e-01,
             8.025606185416310937583009085873554681e-01, 4.362158943603440930655148245148766471e-01,
             1.129272530455059129782111662594436580e-01)
RDPK_C = (0.0, 2.300298624518076223899418286314123354e-01, 4.050046072094990912268498160116125481e-01,
          8.947822893693433545220710894560512805e-01, 7.235136928826589010272834603680114769e-01)
RDPK_BHAT = (1.046363371354093758897668305991705199e-01, 9.520431574956758809511173383346476348e-02,
             4.482446645568668405072421350300379357e-01, 2.449030295461310135957132640369862245e-01,
             1.070116530120251819121660365003405564e-01)
PID_BETA = (0.64, -0.31, 0.04)
PID_ACCEPT_SAFETY = 0.81
RDPK_ORDER_K = 3.0  # min(order, embedded order) + 1


def rdpk3sp35_step(f, u, dt, t=None):
    """One step (t given: f is called as f(u, t + c_i*dt), OrdinaryDiffEq's stage times).  Returns (u_new, utilde) with utilde = dt*sum(bhat_i k_i), so the
    embedded error estimate is (u_new - u) - utilde.

    Register form (3S*+):  tmp=S2, u=S1, uprev=S3:
        tmp += delta_i*u ; u = g1_i*u + g2_i*tmp + g3_i*uprev + beta_i*dt*f(u)
    """
    uprev = u
    k = f(u) if t is None else f(u, t)
    tmp = uprev.copy()
    u = tmp + RDPK_BETA[0] * dt * k
    ut = RDPK_BHAT[0] * dt * k
    for i in range(1, 5):
        k = f(u) if t is None else f(u, t + RDPK_C[i] * dt)
        tmp = tmp + RDPK_DELTA[i] * u
        u = RDPK_G1[i] * u + RDPK_G2[i] * tmp + RDPK_G3[i] * uprev + RDPK_BETA[i] * dt * k
        ut = ut + RDPK_BHAT[i] * dt * k
    return u, ut


def _rms_scaled(err, u0, u1, abstol, reltol):
    """OrdinaryDiffEq calculate_residuals + default internalnorm (RMS)."""
    sk = abstol + np.maximum(np.abs(u0), np.abs(u1)) * reltol
    return math.sqrt(np.mean((err / sk) ** 2))


def initial_dt(f, u0, tspan_len, abstol, reltol, dtmax, order=3, t0=None):
    """Hairer-Wanner starting step as used by OrdinaryDiffEq (ode_determine_initdt).
    t0 given: f is f(u, t) and the second evaluation is at t0 + dt0."""
    if t0 is not None:
        g, tt = f, [t0]
        f = lambda u: g(u, tt[0])
    sk = abstol + np.abs(u0) * reltol
    d0 = math.sqrt(np.mean((u0 / sk) ** 2))
    f0 = f(u0)
    d1 = math.sqrt(np.mean((f0 / sk) ** 2))
    dt0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    dt0 = min(dt0, dtmax, tspan_len)
    if t0 is not None:
        tt[0] = t0 + dt0
    f1 = f(u0 + dt0 * f0)
    d2 = math.sqrt(np.mean(((f1 - f0) / sk) ** 2)) / dt0
    dm = max(d1, d2)
    dt1 = max(1e-6, dt0 * 1e-3) if dm <= 1e-15 else (0.01 / dm) ** (1.0 / (order + 1))
    return min(100.0 * dt0, dt1, dtmax, tspan_len)


@dataclass
class SolveStats:
    naccept: int = 0
    nreject: int = 0
    nrhs: int = 0
    # diagnostics of the adaptive solve (used by the fuzz tests to recognise draws whose outcome round-off decides):
    max_stall: int = 0             # longest run of attempts that did not advance t (rejections, accepted steps with t + h == t)
    min_stop_gap: float = np.inf   # smallest |distance to the next stop - proposed step| / proposed step over all attempts: a
                                   # step that all but reaches its stop (followed by a sliver step and a dozen recovery steps) or
                                   # all but misses being clipped


SOLVE_DIAG = {"max_stall": 0}  # largest SolveStats.max_stall of the adaptive solves since the caller last reset it (fuzz tests)


def solve(
    f: Callable,
    u0,
    tstops: Sequence[float],
    reltol=1e-8,
    abstol=1e-6,
    dtmax=np.inf,
    maxiters=10 ** 6,
    callback: Optional[Callable] = None,
    callback_times: Sequence[float] = (),
    dt0: Optional[float] = None,
    fixed_dt: Optional[float] = None,
    time_dependent: bool = False,
):
    """Adaptive RDPK3Sp35 + PID solve, saving u at every tstop (tstops[0] = t0).

    Restates simulate_iceflow_UDE! (src/simulations/inversions/inversion_utils.jl:551-572:
    ``solve(prob, RDPK3Sp35(); callback, reltol, maxiters, tstops)``) with the
    mass-balance PeriodicCallback of :498-517 passed as ``callback(u, t) -> u`` at
    ``callback_times``.  Returns (snapshots, stats, cb_increments).

    A callback time that is not a tstop is an integrator stop all the same -- PeriodicCallback adds its own tstops
    (step_MB not a multiple of solver.step) -- but the state there is not part of the result (Sleipnir.create_results
    evaluates the solution at `tstops`, inversion_utils.jl:533-538).
    """
    tstops = [float(t) for t in tstops]
    cbt = set(float(t) for t in callback_times)
    saved = set(tstops)
    if callback is not None:
        tstops = sorted(saved | set(t for t in cbt if tstops[0] < t <= tstops[-1]))
    t = tstops[0]
    u = np.array(u0, F, copy=True)
    snaps = [u.copy()]
    cb_inc = {}
    st = SolveStats()
    if fixed_dt is None:
        dt = (initial_dt(f, u, tstops[-1] - tstops[0], abstol, reltol, dtmax, t0=t if time_dependent else None)
              if dt0 is None else dt0)
        st.nrhs += 2
    else:
        dt = fixed_dt
    e1 = e2 = e3 = 1.0
    stall = 0
    for ts in tstops[1:]:
        while t < ts:
            if st.naccept + st.nreject >= maxiters:
                raise RuntimeError("maxiters reached")
            h = min(dt, dtmax)
            t_before = t
            rem = ts - t
            if fixed_dt is None and h > 0.0:
                st.min_stop_gap = min(st.min_stop_gap, abs(rem - h) / h)
            # land exactly on the stop when the step would end within 100 ulp of it
            clipped = h >= rem or abs(rem - h) <= 100.0 * np.finfo(F).eps * abs(t)
            if clipped:
                h = rem
            un, ut = rdpk3sp35_step(f, u, h, t if time_dependent else None)
            st.nrhs += 5
            if fixed_dt is not None:
                u = un
                t = ts if clipped else t + h
                st.naccept += 1
                continue
            err = (un - u) - ut
            EEst = max(_rms_scaled(err, u, un, abstol, reltol), np.finfo(F).eps)
            e1 = 1.0 / EEst
            fac = e1 ** (PID_BETA[0] / RDPK_ORDER_K) * e2 ** (PID_BETA[1] / RDPK_ORDER_K) * e3 ** (PID_BETA[2] / RDPK_ORDER_K)
            fac = 1.0 + math.atan(fac - 1.0)
            if fac >= PID_ACCEPT_SAFETY:
                u = un
                t = ts if clipped else t + h
                e3, e2 = e2, e1
                st.naccept += 1
            else:
                st.nreject += 1
            dt = h * fac
            # a stuck solve (the device's rule, sia2d_device.hpp controller_decide): STALL_MAX attempts in a row without advancing t
            # -- rejections, or accepted steps with t + h == t.  OrdinaryDiffEq aborts earlier, at the first dt <= dtmin = eps(t)
            # (ReturnCode.DtLessThanMin), also where the step size recovers after a few such attempts.
            if t != t_before:
                stall = 0
            else:
                stall += 1
                st.max_stall = max(st.max_stall, stall)
                if stall >= 256:
                    SOLVE_DIAG["max_stall"] = max(SOLVE_DIAG["max_stall"], stall)
                    raise RuntimeError("dtmin: the solve is stuck at t = %r: 256 attempts in a row without advancing t, after %d accepted / %d rejected steps"
                                       % (t, st.naccept, st.nreject))
        if callback is not None and ts in cbt:
            unew = callback(u, ts)
            cb_inc[ts] = unew - u
            u = unew
        if ts in saved:
            snaps.append(u.copy())
    SOLVE_DIAG["max_stall"] = max(SOLVE_DIAG["max_stall"], st.max_stall)
    return snaps, st, cb_inc


def max_diffusivity(H, B, dx, dy, ph: Phys, law: Law, theta=None):
    """max over the dual grid of D(H) (the quantity that bounds an explicit diffusion step)."""
    Hc, S, gSx, gSy, gS, Hbar, ex, ey, exc, eyc = _forward_intermediates(H, B, dx, dy, ph)
    return float(np.max(diffusivity(law, ph, Hbar, gS, theta)))


def solve_euler_cfl(gl: Glacier, law: Law, tstops, cfl=0.25, dtmax=np.inf, theta=None, callback=None,
                    callback_times=()):
    """Explicit Euler with the CFL-limited step dt_n = cfl*min(dx,dy)^2/(4 max D(u_{n-1})) -- the max
    diffusivity of the PREVIOUS state, u_{-1} := u_0 (the device measures max D in the pass that
    advances the state) -- clipped to tstops with the same 100-ulp snapping as `solve`.
    NOT a reference scheme (the reference integrates with OrdinaryDiffEq only): own definition of
    the north star's "CFL" mode, restated here so that the HIP kernel has something to match."""
    tstops = [float(t) for t in tstops]
    cbt = set(float(t) for t in callback_times)
    f = lambda H: sia2d_rhs(H, gl.B, gl.dx, gl.dy, gl.phys, law, theta)
    dmin = min(gl.dx, gl.dy)
    t = tstops[0]
    u = np.array(gl.H0, F, copy=True)
    snaps = [u.copy()]
    nsteps = 0
    Dmax = max_diffusivity(u, gl.B, gl.dx, gl.dy, gl.phys, law, theta)
    for ts in tstops[1:]:
        while t < ts:
            rem = ts - t
            h = cfl * dmin * dmin / (4.0 * Dmax) if Dmax > 0.0 else rem
            h = min(h, dtmax)
            clipped = h >= rem or abs(rem - h) <= 100.0 * np.finfo(F).eps * abs(t)
            if clipped:
                h = rem
            Dnew = max_diffusivity(u, gl.B, gl.dx, gl.dy, gl.phys, law, theta)  # measured on the state being advanced
            u = u + h * f(u)
            Dmax = Dnew
            t = ts if clipped else t + h
            nsteps += 1
        if callback is not None and ts in cbt:
            u = callback(u, ts)
        snaps.append(u.copy())
    return snaps, nsteps


# ----------------------------------------------------------------------------
# Forward simulation of one glacier + loss (batch_loss_iceflow_transient,
# inversion_utils.jl:383-461) and the discrete adjoint (gradient.jl:129-275)
# ----------------------------------------------------------------------------


@dataclass
class SimConfig:
    tstops: Sequence[float]  # includes t0; snapshots saved at each
    reltol: float = 1e-8
    abstol: float = 1e-6
    dtmax: float = np.inf
    mb: Optional[MassBalance] = None
    mb_times: Sequence[float] = ()  # subset of tstops[1:]
    loss_distance: int = 3
    fixed_dt: Optional[float] = None
    # LossDhdt (src/losses/TimeAggregatedLosses.jl:38-113): glacier.dhdtData = (t0, t1, dhdt_ref) with t0, t1 among the
    # stops, and the weight the MultiLoss gives the term (the data loss keeps weight 1)
    dhdt: Optional[Tuple[float, float, float]] = None
    dhdt_weight: float = 1.0
    # LossAvgV (TimeAggregatedLosses.jl:115-258): glacier.velocityData with ONE sample (date1 = t1, date2 = t2) and the
    # MultiLoss weight of the term; every point of t1:step:t2 except the last must be among the stops
    avgv: Optional["AvgVData"] = None
    avgv_weight: float = 1.0
    # VelocityRegularization (src/losses/Regularization.jl:64-79,192-245): Tikhonov penalty on the Laplacian of the predicted
    # surface speed at the velocity-data times `vreg_times` (weights Delta-t.V = their differences), mask =
    # is_in_glacier(H_pred, vreg_distance) & (V > 0), MultiLoss weight vreg_weight (0: off)
    h_log_eps: Optional[float] = None  # LossH's simple loss: None = L2Sum, eps = LogSum(eps) (Losses.jl:34-49)
    vreg_times: Sequence[float] = ()
    vreg_distance: int = 3
    vreg_weight: float = 0.0


def forward(gl: Glacier, law: Law, cfg: SimConfig, theta=None):
    f = lambda H: sia2d_rhs(H, gl.B, gl.dx, gl.dy, gl.phys, law, theta)
    cb = None
    if cfg.mb is not None:
        cb = lambda u, t: mb_apply(cfg.mb, u, gl.B)[0]
    snaps, st, inc = solve(
        f, gl.H0, cfg.tstops, cfg.reltol, cfg.abstol, cfg.dtmax,
        callback=cb, callback_times=cfg.mb_times, fixed_dt=cfg.fixed_dt,
    )
    return snaps, st, inc


def loss_weights(tstops, tH_ref):
    """w_j = tH[m]-tH[m-1] if t_j is the m-th (m>=2) thickness-data time else 0
    (safe_slice, gradient.jl:38-40,144-149; same rule forward: inversion_utils.jl:439-442)."""
    tH = list(tH_ref)
    w = []
    for t in tstops:
        if t in tH:
            m = tH.index(t)
            w.append(tH[m] - tH[m - 1] if m >= 1 else 0.0)
        else:
            w.append(0.0)
    return w


def loss_H(snaps, tstops, H_ref, tH_ref, distance, log_eps=None):
    """sum_tau w_tau * L2Sum(H_tau, Href_tau)/N  (inversion_utils.jl:425-461, Losses.jl:250-270)."""
    w = loss_weights(tstops, tH_ref)
    tH = list(tH_ref)
    N = float(snaps[0].size)
    tot = 0.0
    for j, t in enumerate(tstops):
        if t in tH and w[j] != 0.0:
            Hr = H_ref[tH.index(t)]
            tot += hloss(snaps[j], Hr, is_in_glacier(Hr, distance), N, log_eps) * w[j]
    return tot


def dhdt_loss_terms(snaps, tstops, cfg: SimConfig):
    """LossDhdt, a time-aggregated loss (TimeAggregatedLosses.jl:54-113): with H0, H1 the predictions at dhdtData.t,
    mask = H0 > 1e-2, dhdt = mean((H1 - H0)[mask]) / (t1 - t0):  loss = (dhdt - dhdt_ref)^2, dL/dH0 = -c mask,
    dL/dH1 = +c mask with c = 2 (dhdt - dhdt_ref) / (N_mask (t1 - t0)).  Returns (weighted loss, {stop index: field})."""
    if cfg.dhdt is None:
        return 0.0, {}
    t = [float(x) for x in tstops]
    t0, t1, ref = cfg.dhdt
    i0, i1 = t.index(float(t0)), t.index(float(t1))
    H0, H1 = snaps[i0], snaps[i1]
    mask = H0 > 1e-2
    nm = int(mask.sum())
    dh = float(np.mean(H1[mask] - H0[mask])) / (t1 - t0)
    c = 2.0 * (dh - ref) * mask / (nm * (t1 - t0))
    w = cfg.dhdt_weight
    return w * (dh - ref) ** 2, {i0: -w * c, i1: w * c}


@dataclass
class AvgVData:
    """The single velocity sample LossAvgV compares with (velocityData.date1/date2/vabs/vx/vy, nx*ny arrays with the
    inn1 pairing of V_from_H) and the fields of the loss type (component, step)."""

    t1: float
    t2: float
    Vabs: np.ndarray
    Vx: np.ndarray
    Vy: np.ndarray
    component: str = "xy"
    step: float = 1.0 / 12.0


def avgv_times(a: AvgVData):
    """tLoss = collect(t1:step:t2), dt = diff(tLoss), tLoss[begin:end-1]  (TimeAggregatedLosses.jl:201-204)."""
    # length of a Julia float range: the nearest integer to (t2 - t1) / step, one less when that point overshoots t2 by more
    # than rounding
    n = int(round((a.t2 - a.t1) / a.step))
    if a.t1 + n * a.step > a.t2 + 4.0 * np.finfo(F).eps * max(abs(a.t1), abs(a.t2)):
        n -= 1
    tl = [a.t1 + i * a.step for i in range(n + 1)]
    dt = [tl[i + 1] - tl[i] for i in range(n)]
    return tl[:-1], dt


def _stop_index(t, x):
    j = int(np.argmin(np.abs(np.asarray(t, F) - x)))
    if abs(t[j] - x) > 1e-9:
        raise ValueError(f"time {x} of a time-aggregated loss is not among the stops")  # indFromT
    return j


def avgv_loss_terms(snaps, tstops, cfg: SimConfig, gl, law: Law, theta=None):
    """LossAvgV (TimeAggregatedLosses.jl:146-258): V_i = V_from_H(H(tLoss_i)), time average with weights dt_i / T,
    L2Sum against the single reference sample on mask = V_ref > 0 (component :xy or :abs; normalization = prod(N));
    cotangents dl/dV dt_i / T pulled back through surface_V at every tLoss_i (VJP_lambda_dsurface_V/dH and /dtheta).
    Returns (weighted loss, {stop index: dL/dH field}, dL/dtheta)."""
    P = 1 if law.kind == LAW_CONST_A else law.mlp.n_params
    a = cfg.avgv
    if a is None:
        return 0.0, {}, np.zeros(P)
    t = [float(x) for x in tstops]
    tl, dt = avgv_times(a)
    T = sum(dt)
    idx = [_stop_index(t, x) for x in tl]
    N = float(gl.B.size)
    ph = gl.phys
    avx = np.zeros_like(gl.B)
    avy = np.zeros_like(gl.B)
    for i, j in enumerate(idx):
        Vx, Vy, _ = V_from_H(snaps[j], gl.B, gl.dx, gl.dy, ph, law, theta)
        avx = avx + (Vx * dt[i]) / T
        avy = avy + (Vy * dt[i]) / T
    av = np.sqrt(avx ** 2 + avy ** 2)
    mask = a.Vabs > 0.0
    if a.component == "xy":
        l = l2sum_loss(avx, a.Vx, mask, N) + l2sum_loss(avy, a.Vy, mask, N)
        dVx = l2sum_backward(avx, a.Vx, mask, N)
        dVy = l2sum_backward(avy, a.Vy, mask, N)
    elif a.component == "abs":
        l = l2sum_loss(av, a.Vabs, mask, N)
        dV = l2sum_backward(av, a.Vabs, mask, N)
        with np.errstate(divide="ignore", invalid="ignore"):
            dVx = np.where(mask, dV * (avx - a.Vx) / (av - a.Vabs), 0.0)
            dVy = np.where(mask, dV * (avy - a.Vy) / (av - a.Vabs), 0.0)
    else:
        raise ValueError("Loss type not implemented.")
    w = cfg.avgv_weight
    dl, dth = {}, np.zeros(P)
    for i, j in enumerate(idx):
        cx, cy = dVx * (dt[i] / T), dVy * (dt[i] / T)
        dl[j] = w * vjp_surface_V_H(cx, cy, snaps[j], gl.B, gl.dx, gl.dy, ph, law, theta)
        dth = dth + w * vjp_surface_V_theta(cx, cy, snaps[j], gl.B, gl.dx, gl.dy, ph, law, theta)
    return w * l, dl, dth


def vreg_backward(H, gl, law: Law, theta, distance):
    """VelocityRegularization(reg = TikhonovRegularization(), components = :abs) at one state, Delta-t = 1
    (Regularization.jl:192-245): loss = sum_mask (lap V)^2 with V = |V_from_H(H)|, mask = is_in_glacier(H, distance) & (V > 0);
    dReg/dV = VJP_lap(2 mask lap V) (:110-126), dReg/dVx = dReg/dV Vx / V where V > 0, pulled back through surface_V.
    Returns (loss, dL/dH, dL/dtheta)."""
    ph = gl.phys
    Vx, Vy, V = V_from_H(H, gl.B, gl.dx, gl.dy, ph, law, theta)
    mask = is_in_glacier(H, distance) & (V > 0.0)
    lap = laplacian(V, gl.dx, gl.dy)
    c = np.zeros_like(V)
    c[mask] = 2.0 * lap[mask]
    dV = vjp_laplacian(c, gl.dx, gl.dy)
    with np.errstate(divide="ignore", invalid="ignore"):
        dVx = np.where(V > 0.0, dV * Vx / V, 0.0)
        dVy = np.where(V > 0.0, dV * Vy / V, 0.0)
    return (float(np.sum(lap[mask] ** 2)), vjp_surface_V_H(dVx, dVy, H, gl.B, gl.dx, gl.dy, ph, law, theta),
            vjp_surface_V_theta(dVx, dVy, H, gl.B, gl.dx, gl.dy, ph, law, theta))


def vreg_loss_terms(snaps, tstops, cfg: SimConfig, gl, law: Law, theta=None, quadrature=None):
    """The VelocityRegularization term of a MultiLoss over a run: loss and dL/dH at the velocity-data stops with the weights
    Delta-t.V of the discrete loss (gradient.jl:144-163, :331-365 for the reverse ODE); dL/dtheta summed over the same stops
    (DiscreteAdjoint, :252) or -- quadrature = (nodes, weights) -- integrated over the Gauss-Legendre nodes on the
    interpolated state with Delta-t = 1 (ContinuousAdjoint, :475-503).  Returns (loss, {stop: dL/dH}, dL/dtheta)."""
    P = 1 if law.kind == LAW_CONST_A else law.mlp.n_params
    if not (cfg.vreg_weight != 0.0) or len(cfg.vreg_times) == 0:
        return 0.0, {}, np.zeros(P)
    t = [float(x) for x in tstops]
    w = loss_weights(t, cfg.vreg_times)
    lam_w = cfg.vreg_weight
    l, dl, dth = 0.0, {}, np.zeros(P)
    for j in range(len(t)):
        if w[j] == 0.0:
            continue
        lj, gH, gth = vreg_backward(snaps[j], gl, law, theta, cfg.vreg_distance)
        l += lam_w * w[j] * lj
        dl[j] = lam_w * w[j] * gH
        if quadrature is None:
            dth = dth + lam_w * w[j] * gth
    if quadrature is not None:
        for tn, wn in zip(*quadrature):
            _, _, gth = vreg_backward(linear_itp(t, snaps, float(tn)), gl, law, theta, cfg.vreg_distance)
            dth = dth + lam_w * wn * gth
    return l, dl, dth


def aggregated_loss_terms(snaps, tstops, cfg: SimConfig, gl, law: Law, theta=None, quadrature=None):
    """Every term of the loss whose gradient does not involve lambda and is therefore formed right after the forward solve:
    the time-aggregated losses (MultiLoss of LossDhdt / LossAvgV; TimeAggregatedLosses.jl:292-345) and
    VelocityRegularization.  (loss, {stop index: dL/dH}, dL/dtheta)."""
    l1, d1 = dhdt_loss_terms(snaps, tstops, cfg)
    l2, d2, th2 = avgv_loss_terms(snaps, tstops, cfg, gl, law, theta)
    l3, d3, th3 = vreg_loss_terms(snaps, tstops, cfg, gl, law, theta, quadrature)
    d = dict(d1)
    for dd in (d2, d3):
        for j, f in dd.items():
            d[j] = d[j] + f if j in d else f
    return l1 + l2 + l3, d, th2 + th3


def _vjp_H_of(vjp):
    """VJP_lambda_dSIA/dH dispatch on the VJP method (VJPs.jl:2-10)."""
    return {"discrete": vjp_H, "continuous": vjp_H_continuous}[vjp]


def loss_and_grad(gl: Glacier, law: Law, cfg: SimConfig, H_ref, tH_ref, theta=None, vjp="discrete"):
    """SIA2D_grad_batch! with DiscreteAdjoint + DiscreteVJP | ContinuousVJP (gradient.jl:45-275).

    Reverse loop (explicit Euler over the snapshots, Jacobian at the END-of-interval
    state, theta-VJP with the already-updated lambda) exactly as :191-253.
    Returns (loss, dL/dtheta, lambda_0)."""
    snaps, st, inc = forward(gl, law, cfg, theta)
    t = list(cfg.tstops)
    k = len(t)
    N = float(gl.B.size)
    w = loss_weights(t, tH_ref)
    tH = list(tH_ref)
    P = 1 if law.kind == LAW_CONST_A else law.mlp.n_params
    dLdtheta = np.zeros(P)
    lam = [np.zeros_like(gl.B) for _ in range(k)]
    loss_rev = 0.0
    l_agg, dl_agg, dth_agg = aggregated_loss_terms(snaps, t, cfg, gl, law, theta)  # gradient.jl:170-188
    for j in reversed(range(k)):
        tj = t[j]
        if cfg.mb is not None and tj in cfg.mb_times:  # :201-207
            H_pre = snaps[j] - inc[tj]
            lam[j] = lam[j] + vjp_mb(cfg.mb, lam[j], H_pre, gl.B)
        if tj in tH and w[j] != 0.0:
            Hr = H_ref[tH.index(tj)]
            mask = is_in_glacier(Hr, cfg.loss_distance)
            dl = hloss_backward(snaps[j], Hr, mask, N, cfg.h_log_eps) * w[j]
            loss_rev += hloss(snaps[j], Hr, mask, N, cfg.h_log_eps) * w[j]
        else:
            dl = 0.0
        if j in dl_agg:  # :212-215 (like every dl/dH of the first stop, the j = 0 term is never used: there is no lambda[-1])
            dl = dl + dl_agg[j]
        g = _vjp_H_of(vjp)(lam[j], snaps[j], gl.B, gl.dx, gl.dy, gl.phys, law, theta)  # :235-237
        if j > 0:
            dt = t[j] - t[j - 1]
            lam[j - 1] = lam[j] + dt * g + dl  # :242
            dth = vjp_theta(lam[j - 1], snaps[j], gl.B, gl.dx, gl.dy, gl.phys, law, theta)  # :245-246
            dLdtheta += dt * dth  # :249
    loss_fwd = loss_H(snaps, t, H_ref, tH_ref, cfg.loss_distance, cfg.h_log_eps)
    assert math.isclose(loss_rev, loss_fwd, rel_tol=1e-8, abs_tol=0.0) or loss_fwd == 0.0  # :259
    return loss_fwd + l_agg, dLdtheta + dth_agg, lam[0]  # :254-255, :274


# ----------------------------------------------------------------------------
# Continuous adjoint (gradient.jl:276-539; defaults AdjointTypes.jl:53-67) with the
# DiscreteVJP stencils: reverse ODE  dlam/dtau = J_H(H_itp(-tau))^T lam  solved with the same
# adaptive RDPK3Sp35, loss / mass-balance contributions as callbacks at the snapshot times,
# dL/dtheta by Gauss-Legendre quadrature of  J_theta(H_itp(t))^T lam(t).
# ----------------------------------------------------------------------------


@dataclass
class ContinuousAdjointCfg:  # AdjointTypes.jl:58-67
    reltol: float = 1e-8
    abstol: float = 1e-8
    dtmax: float = 1.0 / 12.0
    n_quadrature: int = 200
    maxiters: int = 10 ** 6


def gauss_quadrature(t0, t1, n):
    """GaussQuadrature (gradient.jl:560-566): Gauss-Legendre nodes / weights mapped to [t0,t1]."""
    x, w = np.polynomial.legendre.leggauss(n)
    return (t0 + t1) / 2.0 + x * (t1 - t0) / 2.0, (t1 - t0) / 2.0 * w


def linear_itp(ts, fields, t):
    """interpolate((t,), H, Gridded(Linear())) (gradient.jl:287)."""
    k = len(ts)
    j = min(max(int(np.searchsorted(ts, t, side="right")) - 1, 0), k - 2)
    s = (t - ts[j]) / (ts[j + 1] - ts[j])
    if s == 0.0:
        return fields[j]
    if s == 1.0:
        return fields[j + 1]
    return (1.0 - s) * fields[j] + s * fields[j + 1]


def loss_and_grad_continuous(gl: Glacier, law: Law, cfg: SimConfig, H_ref, tH_ref, adj: ContinuousAdjointCfg = None,
                             theta=None, vjp="discrete", V_ref=None, tV_ref=(), vspec=None, loss_kind="H", scaling=1.0):
    """SIA2D_grad_batch! with ContinuousAdjoint (gradient.jl:276-539) for LossH, LossV or LossHV.
    Returns (loss, dL/dtheta, lambda(t0), stats of the reverse solve).

    Thickness / velocity terms enter lambda at their data times with the Delta-t weights of the discrete
    loss (:331-365); the explicit theta-dependence of the velocity loss is integrated by the quadrature
    with Delta-t = 1 and the reference velocities interpolated linearly in time (:475-503, :291-301) --
    which needs velocity data spanning tspan (Gridded(Linear()) does not extrapolate) unless there is a
    single velocity map (constant interpolator)."""
    adj = adj or ContinuousAdjointCfg()
    snaps, st, inc = forward(gl, law, cfg, theta)
    t = [float(x) for x in cfg.tstops]
    k = len(t)
    N = float(gl.B.size)
    useH, useV = loss_kind in ("H", "HV"), loss_kind in ("V", "HV")
    dtH = loss_weights(t, tH_ref) if useH else [0.0] * k
    dtV = loss_weights(t, tV_ref) if useV else [0.0] * k
    wH = [d * d if loss_kind == "HV" else d for d in dtH]          # Losses.jl:407,424-431
    wV = [scaling * d * d if loss_kind == "HV" else d for d in dtV]
    wq = scaling if loss_kind == "HV" else 1.0                     # quadrature: Delta-t = (1, 1)  (:474)
    tH = [float(x) for x in tH_ref]
    tV = [float(x) for x in tV_ref]
    mbt = set(float(x) for x in cfg.mb_times) if cfg.mb is not None else set()
    H_itp = lambda tt: linear_itp(t, snaps, tt)  # :287

    def V_itp(tt):  # :291-301
        if len(tV) == 1:
            return V_ref[0]
        if not (tV[0] <= tt <= tV[-1]):
            raise ValueError("velocity data must span tspan (Gridded(Linear()) does not extrapolate)")
        return tuple(linear_itp(tV, [v[c] for v in V_ref], tt) for c in range(3))

    def effect_loss(tt, u):  # :331-365; dt weights included, first data time has weight 0
        j = t.index(tt)
        if wH[j] != 0.0:
            Hr = H_ref[tH.index(tt)]
            u = u + hloss_backward(H_itp(tt), Hr, is_in_glacier(Hr, cfg.loss_distance), N, cfg.h_log_eps) * wH[j]
        if wV[j] != 0.0:
            Va, Vxr, Vyr = V_ref[tV.index(tt)]
            gH, _ = backward_loss_V(vspec, H_itp(tt), gl.B, gl.dx, gl.dy, gl.phys, law, Va, Vxr, Vyr, N, theta)
            u = u + gH * wV[j]
        return u

    def effect_mb(tt, u):  # :413-425
        if tt in mbt:
            return u + vjp_mb(cfg.mb, u, H_itp(tt) - inc[tt], gl.B)
        return u

    nodes, wts = gauss_quadrature(t[0], t[-1], adj.n_quadrature)  # :307-308
    l_agg, dl_agg, dth_agg = aggregated_loss_terms(snaps, t, cfg, gl, law, theta, (nodes, wts))  # :369-387

    def effect_agg(tt, u):  # :389-399
        j = t.index(tt)
        return u + dl_agg[j] if j in dl_agg else u

    f_rev = lambda lam, tau: _vjp_H_of(vjp)(lam, H_itp(-tau), gl.B, gl.dx, gl.dy, gl.phys, law, theta)  # :316-324
    lam1 = effect_loss(t[-1], np.zeros_like(gl.B))  # :441-446 (not covered by the discrete callback)
    lam1 = effect_agg(t[-1], lam1)  # :447-449
    lam1 = effect_mb(t[-1], lam1)  # PeriodicCallback(initial_affect = true) :431-432
    snap_tau = [-x for x in reversed(t)]
    stops = sorted(set(snap_tau) | set(-float(x) for x in nodes))  # :457
    tset = set(t)
    # mass-balance times that are not result stops (step_MB not a multiple of solver.step): the reverse PeriodicCallback
    # (:426-432) stops the integrator there too and adds VJP_MB(lambda, H_itp(t) - MB_t) -- H_itp interpolates the RESULT
    mb_only = sorted(-x for x in mbt if x not in tset and t[0] < x < t[-1])

    def cb(u, tau):  # CallbackSet order: MB, loss, aggregated :437
        tt = -tau
        if tt in tset:
            return effect_agg(tt, effect_loss(tt, effect_mb(tt, u)))
        return effect_mb(tt, u)

    lam_s, st_rev, _ = solve(f_rev, lam1, stops, adj.reltol, adj.abstol, adj.dtmax, adj.maxiters,
                             callback=cb, callback_times=snap_tau[1:] + mb_only, time_dependent=True)
    P = 1 if law.kind == LAW_CONST_A else law.mlp.n_params
    dLdtheta = np.zeros(P)
    at = {tau: i for i, tau in enumerate(stops)}
    for tn, wn in zip(nodes, wts):  # :497-503
        tn = float(tn)
        lam = lam_s[at[-tn]]
        Hn = H_itp(tn)
        g = vjp_theta(lam, Hn, gl.B, gl.dx, gl.dy, gl.phys, law, theta)
        if useV:
            Va, Vxr, Vyr = V_itp(tn)
            _, gth = backward_loss_V(vspec, Hn, gl.B, gl.dx, gl.dy, gl.phys, law, Va, Vxr, Vyr, N, theta)
            g = g + wq * gth
        dLdtheta += wn * g
    loss = 0.0
    if useH:
        for j in range(k):
            if wH[j] != 0.0:
                Hr = H_ref[tH.index(t[j])]
                loss += hloss(snaps[j], Hr, is_in_glacier(Hr, cfg.loss_distance), N, cfg.h_log_eps) * wH[j]
    if useV:
        for j in range(k):
            if wV[j] != 0.0:
                Va, Vxr, Vyr = V_ref[tV.index(t[j])]
                loss += loss_V(vspec, snaps[j], gl.B, gl.dx, gl.dy, gl.phys, law, Va, Vxr, Vyr, N, theta) * wV[j]
    return loss + l_agg, dLdtheta + dth_agg, lam_s[-1], st_rev  # :538


# ----------------------------------------------------------------------------
# Tikhonov regularisation (src/losses/Regularization.jl:92-126, 330-382) and the
# initial-condition filters (src/models/trainable_components/InitialCondition_utils.jl:30-141)
# ----------------------------------------------------------------------------


def laplacian(a, dx, dy):
    """nabla^2 (Regularization.jl:330-352): staggered second differences averaged back to the primal
    interior; 0 on the boundary ring."""
    d2x = avg_y(diff_x(avg_y(diff_x(a) / dx)) / dx)
    d2y = avg_x(diff_y(avg_x(diff_y(a) / dy)) / dy)
    out = np.zeros_like(a)
    out[1:-1, 1:-1] = d2x + d2y
    return out


def vjp_laplacian(lam, dx, dy):
    """VJP_lambda_dnabla2a_da (Regularization.jl:372-382)."""
    li = lam[1:-1, 1:-1]
    gx = diff_x_adjoint(avg_y_adjoint(diff_x_adjoint(avg_y_adjoint(li), dx)), dx)
    gy = diff_y_adjoint(avg_x_adjoint(diff_y_adjoint(avg_x_adjoint(li), dy)), dy)
    return gx + gy


def tikhonov_loss(a, dx, dy, mask):
    """loss(::TikhonovRegularization) (Regularization.jl:92-101): sum over mask of (nabla^2 a)^2."""
    return float(np.sum(laplacian(a, dx, dy)[mask] ** 2))


def tikhonov_backward(a, dx, dy, mask):
    """backward_loss(::TikhonovRegularization) (Regularization.jl:102-115)."""
    g = np.zeros_like(a)
    g[mask] = 2.0 * laplacian(a, dx, dy)[mask]
    return vjp_laplacian(g, dx, dy)


def sigma_zang(x, beta=2.0):
    """InitialCondition_utils.jl:92-100."""
    return np.where(x < -beta / 2, 0.0, np.where(x < beta / 2, (x + beta / 2) ** 2 / (2 * beta), x))


def dsigma_zang(x, beta=2.0):
    """InitialCondition_utils.jl:112-120."""
    return np.where(x < -beta / 2, 0.0, np.where(x < beta / 2, x / beta + 0.5, 1.0))


def evaluate_H0(theta_ic, outside_mask, filt="identity"):
    """evaluate_H0 (InitialCondition_utils.jl:30-46): filter, then zero outside the glacier."""
    if filt == "identity":
        H0 = np.array(theta_ic, F, copy=True)
    elif filt == "softplus":
        H0 = np.log(1.0 + np.exp(theta_ic))
    elif filt == "Zang1980":
        H0 = sigma_zang(np.asarray(theta_ic, F))
    else:
        raise ValueError(filt)
    H0 = np.array(H0, F)
    H0[outside_mask] = 0.0
    return H0


def evaluate_dH0(theta_ic, outside_mask, filt="identity"):
    """evaluate_dH0 (InitialCondition_utils.jl:73-89)."""
    if filt == "identity":
        d = np.ones_like(np.asarray(theta_ic, F))
    elif filt == "softplus":
        d = 1.0 / (1.0 + np.exp(-np.asarray(theta_ic, F)))
    elif filt == "Zang1980":
        d = dsigma_zang(np.asarray(theta_ic, F))
    else:
        raise ValueError(filt)
    d = np.array(d, F)
    d[outside_mask] = 0.0
    return d


# ----------------------------------------------------------------------------
# Known answer: Halfar (1983) similarity solution, n=3, flat bed, no MB
# (reference set-up test/test_grad_loss.jl:526-539)
# ----------------------------------------------------------------------------


def halfar_t0(A, h0, r0, rho=900.0, g=9.81, n=3.0):
    Gam = 2.0 * A * (rho * g) ** n / (n + 2.0)
    return (1.0 / (18.0 * Gam)) * (7.0 / 4.0) ** 3 * r0 ** 4 / h0 ** 7


def halfar(x, y, t, A, h0, r0, rho=900.0, g=9.81):
    """H(r,t), t absolute (dome has height h0, radius r0 at t = t0)."""
    t0 = halfar_t0(A, h0, r0, rho, g)
    r = np.sqrt(x ** 2 + y ** 2)
    s = (t0 / t) ** (1.0 / 9.0)
    br = 1.0 - ((t0 / t) ** (1.0 / 18.0) * r / r0) ** (4.0 / 3.0)
    return np.where(br > 0.0, h0 * s * np.maximum(br, 0.0) ** (3.0 / 7.0), 0.0)


# ----------------------------------------------------------------------------
# Synthetic inputs shared by tests / bench (seed 1234)
# ----------------------------------------------------------------------------


def synthetic_icecap(nx, ny, dx=100.0, seed=1234, bumpy=True):
    """512^2-style ice cap: B = 500+50 sin cos + 0.01 x, H0 = max(0, 800(1-(r/R)^2))."""
    x = (np.arange(nx) * dx)[:, None]
    y = (np.arange(ny) * dx)[None, :]
    Lx, Ly = nx * dx, ny * dx
    B = 500.0 + 50.0 * np.sin(2 * np.pi * x / Lx) * np.cos(2 * np.pi * y / Ly) + 0.01 * x
    if not bumpy:
        B = np.zeros((nx, ny)) + 0.0 * x
    r = np.sqrt((x - Lx / 2) ** 2 + (y - Ly / 2) ** 2)
    R = 0.4 * min(Lx, Ly)
    H0 = np.maximum(0.0, 800.0 * (1.0 - (r / R) ** 2))
    return np.asfortranarray(H0), np.asfortranarray(B + 0.0 * H0)


def synthetic_valley(nx, ny, dx=50.0):
    """Argentiere stand-in: sloping parabolic valley with a tongue of ice."""
    x = (np.arange(nx) * dx)[:, None]
    y = (np.arange(ny) * dx)[None, :]
    yc, yh = ny * dx / 2, ny * dx / 2
    B = 2200.0 - 0.12 * x + 300.0 * ((y - yc) / yh) ** 2
    ell = ((x - 0.45 * nx * dx) / (0.38 * nx * dx)) ** 2 + ((y - yc) / (0.30 * ny * dx)) ** 2
    H0 = np.maximum(0.0, 250.0 * (1.0 - ell))
    return np.asfortranarray(H0), np.asfortranarray(B + 0.0 * H0)


def synthetic_alpine(nx, ny, dx=50.0, hmax=110.0, slope=0.08):
    """Gentle valley glacier with alpine-like dynamics (metres of thickness change per year):
    stand-in for the README's RGI glaciers (BASELINE configs[3]), whose data cannot be
    downloaded here.  Slow enough that the reference's monthly explicit-Euler adjoint is
    accurate (ratio ~3e-4 vs finite differences over 2 years)."""
    x = (np.arange(nx) * dx)[:, None]
    y = (np.arange(ny) * dx)[None, :]
    yc = ny * dx / 2
    B = 2200.0 - slope * x + 300.0 * ((y - yc) / yc) ** 2
    ell = ((x - 0.45 * nx * dx) / (0.38 * nx * dx)) ** 2 + ((y - yc) / (0.30 * ny * dx)) ** 2
    H0 = np.maximum(0.0, hmax * (1.0 - ell))
    return np.asfortranarray(H0), np.asfortranarray(B + 0.0 * H0)


# ----------------------------------------------------------------------------
# Surface-velocity path: Huginn.surface_V / V_from_H are out of tree; the
# forward is what the reference's discrete VJPs re-execute (adjoint.jl:268-413) and what its
# test differentiates (test/SIA2D_adjoint.jl:209-330:  <Vx, inn1(w1)> + <Vy, inn1(w2)>).
# Target :A only (the reference's Velocity^ family for the other targets is "not correct"
# by its own comment, target_D_pure.jl).
# ----------------------------------------------------------------------------


def gamma_up_no_A(ph: Phys):
    """Gamma^(...; include_A=false) = 2 (rho g)^n/(n+1)  (target_utils.jl:20-29)."""
    return 2.0 * (ph.rho * ph.g) ** ph.n / (ph.n + 1.0)


def inn1(a):
    """Huginn.inn1: A[1:end-1, 1:end-1] (pairs the dual grid with an nx*ny array)."""
    return a[:-1, :-1]


def _A_dual(law: Law, ph: Phys, theta=None):
    if law.kind not in (LAW_CONST_A, LAW_NN_A_SCALAR, LAW_NN_A_GRIDDED):
        raise ValueError("surface velocity is provided for the :A target only")
    return law_value(law, ph, None, None, theta)


def _velocity_up_hybrid(law: Law, ph: Phys, Hbar, gradS, Y):
    """compute_Velocity^ of target :D_hybrid AS WRITTEN (target_D_hybrid.jl:353-372): the sliding term of the diffusivity
    and Y Gamma H^(n_H+1) |grad S|^(n_gradS-1) with Gamma = 2 (rho g)^n / (n + 2) -- the DIFFUSIVITY's Gamma (the variable is
    named Gamma^_no_A but assigned Gamma(...), :362), not Gamma^ = 2 (rho g)^n / (n + 1)."""
    nH, nS = _hyb_exps(law, ph)
    D = Y * gamma_no_A(ph) * _pow(Hbar, nH + 1.0) * _pow(gradS, nS - 1.0)
    Sc = sliding_S(ph)
    if Sc != 0.0:
        D = D + Sc * _pow(Hbar, ph.p - ph.q + 1.0) * _pow(gradS, ph.p - 1.0)
    return D


def _diffusivity_hybrid_with_Y(law: Law, ph: Phys, Hbar, gradS, Y):
    """compute_D(target, Y; ...) of target :D_hybrid (target_D_hybrid.jl:190-208)."""
    nH, nS = _hyb_exps(law, ph)
    D = Y * gamma_no_A(ph) * _pow(Hbar, nH + 2.0) * _pow(gradS, nS - 1.0)
    Sc = sliding_S(ph)
    if Sc != 0.0:
        D = Sc * _pow(Hbar, ph.p - ph.q + 1.0) * _pow(gradS, ph.p - 1.0) + D
    return D


def velocity_up(law: Law, ph: Phys, Hbar, gradS, theta=None):
    """Velocity^ (target_A.jl:94-108), sliding term exactly as written there; target :D: U / f (target_D_pure.jl:206-217);
    target :D_hybrid: _velocity_up_hybrid."""
    if law.kind == LAW_NN_U:
        return law_value(law, ph, Hbar, gradS, theta) / law.fV
    if law.kind == LAW_NN_Y:
        return _velocity_up_hybrid(law, ph, Hbar, gradS, law_value(law, ph, Hbar, gradS, theta))
    A = _A_dual(law, ph, theta)
    D = A * gamma_up_no_A(ph) * _pow(Hbar, ph.n + 1.0) * _pow(gradS, ph.n - 1.0)
    Sc = sliding_S(ph)
    if Sc != 0.0:
        D = D + Sc * (ph.p - ph.q + 2.0) * _pow(Hbar, ph.p - ph.q + 1.0) * _pow(gradS, ph.n - 1.0)
    return D


def d_velocity_up_dH(law: Law, ph: Phys, Hbar, gradS, theta=None):
    """dVelocity^/dH (target_A.jl:110-125); target :D: central difference of U with step 1e-4 (target_D_pure.jl:219-231)."""
    if law.kind == LAW_NN_U:
        d = 1e-4
        return (1.0 / law.fV) * (law_value(law, ph, Hbar + d, gradS, theta) - law_value(law, ph, Hbar - d, gradS, theta)) / (2.0 * d)
    if law.kind == LAW_NN_Y:
        # dVelocity^/dH of target :D_hybrid AS WRITTEN (target_D_hybrid.jl:226-262): the closed-form part differentiates
        # Y Gamma H^(n_H+1) and the sliding term; the network part is a forward difference (1e-4) of compute_D -- the
        # DIFFUSIVITY (H^(n_H+2)), not of compute_Velocity^ -- with Y evaluated at Hbar + dH and at Hbar.
        nH, nS = _hyb_exps(law, ph)
        Y = law_value(law, ph, Hbar, gradS, theta)
        out = (nH + 1.0) * Y * gamma_no_A(ph) * _pow(Hbar, nH) * _pow(gradS, nS - 1.0)
        Sc = sliding_S(ph)
        if Sc != 0.0:
            out = (ph.p - ph.q + 1.0) * Sc * _pow(Hbar, ph.p - ph.q) * _pow(gradS, ph.p - 1.0) + out
        d = 1e-4
        a = _diffusivity_hybrid_with_Y(law, ph, Hbar, gradS, law_value(law, ph, Hbar + d, gradS, theta))
        b = _diffusivity_hybrid_with_Y(law, ph, Hbar, gradS, Y)
        return out + (a - b) / d
    A = _A_dual(law, ph, theta)
    out = A * gamma_up_no_A(ph) * (ph.n + 1.0) * _pow(Hbar, ph.n) * _pow(gradS, ph.n - 1.0)
    Sc = sliding_S(ph)
    if Sc != 0.0:
        out = out + Sc * (ph.p - ph.q + 2.0) * _pow(Hbar, ph.p - ph.q) * _pow(gradS, ph.n - 1.0)
    return out


def d_velocity_up_dgradS(law: Law, ph: Phys, Hbar, gradS, theta=None):
    """dVelocity^/dgradH (target_A.jl:127-142); target :D: central difference with step 1e-6, not divided by |grad S|
    (target_D_pure.jl:233-245)."""
    if law.kind == LAW_NN_U:
        d = 1e-6
        return (1.0 / law.fV) * (law_value(law, ph, Hbar, gradS + d, theta) - law_value(law, ph, Hbar, gradS - d, theta)) / (2.0 * d)
    if law.kind == LAW_NN_Y:
        # AS WRITTEN (target_D_hybrid.jl:264-285): Gamma^ = 2 (rho g)^n / (n + 1) and H^(n_H+2) here
        nH, nS = _hyb_exps(law, ph)
        out = gamma_up_no_A(ph) * law_value(law, ph, Hbar, gradS, theta) * (nS - 1.0) * _pow(Hbar, nH + 2.0) * _pow(gradS, nS - 3.0)
        Sc = sliding_S(ph)
        if Sc != 0.0:
            out = (ph.p - 1.0) * Sc * _pow(Hbar, ph.p - ph.q + 1.0) * _pow(gradS, ph.p - 3.0) + out
        return out
    A = _A_dual(law, ph, theta)
    out = A * gamma_up_no_A(ph) * (ph.n - 1.0) * _pow(Hbar, ph.n + 1.0) * _pow(gradS, ph.n - 3.0)
    Sc = sliding_S(ph)
    if Sc != 0.0:
        out = out + Sc * (ph.p - ph.q + 2.0) * (ph.p - 1.0) * _pow(Hbar, ph.p - ph.q + 1.0) * _pow(gradS, ph.n - 3.0)
    return out


def surface_V(H, B, dx, dy, ph: Phys, law: Law, theta=None):
    """(Vx, Vy) on the dual grid: V = -Velocity^ * grad S."""
    Hc, S, gSx, gSy, gS, Hbar, *_ = _forward_intermediates(H, B, dx, dy, ph)
    D = velocity_up(law, ph, Hbar, gS, theta)
    return -D * gSx, -D * gSy


def V_from_H(H, B, dx, dy, ph: Phys, law: Law, theta=None):
    """(Vx, Vy, V) as nx*ny arrays with inn1(.) = surface_V, 0 on the last row / column."""
    vx, vy = surface_V(H, B, dx, dy, ph, law, theta)
    Vx = np.zeros_like(H)
    Vy = np.zeros_like(H)
    Vx[:-1, :-1] = vx
    Vy[:-1, :-1] = vy
    return Vx, Vy, np.sqrt(Vx ** 2 + Vy ** 2)


def vjp_surface_V_H(dVx, dVy, H, B, dx, dy, ph: Phys, law: Law, theta=None):
    """VJP_lambda_dsurface_V/dH_discrete (adjoint.jl:268-350).  dVx, dVy are nx*ny."""
    Hc, S, gSx, gSy, gS, Hbar, *_ = _forward_intermediates(H, B, dx, dy, ph)
    alpha = d_velocity_up_dH(law, ph, Hbar, gS, theta)
    beta = d_velocity_up_dgradS(law, ph, Hbar, gS, theta)
    wx, wy = inn1(dVx), inn1(dVy)
    gSdV = gSx * wx + gSy * wy
    dDdH = (avg_adjoint(alpha * gSdV) + diff_x_adjoint(avg_y_adjoint(beta * gSx * gSdV), dx)
            + diff_y_adjoint(avg_x_adjoint(beta * gSy * gSdV), dy))
    D = velocity_up(law, ph, Hbar, gS, theta)
    dgS = diff_x_adjoint(avg_y_adjoint(D * wx), dx) + diff_y_adjoint(avg_x_adjoint(D * wy), dy)
    return -(dDdH + dgS)


def vjp_surface_V_theta(dVx, dVy, H, B, dx, dy, ph: Phys, law: Law, theta=None):
    """VJP_lambda_dsurface_V/dtheta_discrete (adjoint.jl:352-413)."""
    Hc, S, gSx, gSy, gS, Hbar, *_ = _forward_intermediates(H, B, dx, dy, ph)
    gSdV = gSx * inn1(dVx) + gSy * inn1(dVy)
    if law.kind == LAW_NN_U:
        # dVelocity^/dtheta = dU/dtheta / f (target_D_pure.jl:247-255) with dU/dtheta = (Hbar > 0) x the law gradient, exact per
        # node (:None, :163-176) or interpolated bilinearly on LawU's node grid (:Linear, :179-193) -- target.interpolation
        kind, nhalf = law.interp()
        if kind == "linear":
            g = law_grad_theta_bilinear(law, ph, Hbar, gS, theta, nhalf)
        else:
            g = law_grad_theta(law, ph, Hbar, gS, theta)
        return -np.tensordot(g, (Hbar > 0.0) * gSdV / law.fV, axes=([1, 2], [0, 1]))
    if law.kind == LAW_NN_Y:
        # dVelocity^/dtheta of target :D_hybrid (target_D_hybrid.jl:287-351): Gamma^ H^(n_H+1) |grad S|^(n_gradS-1) x dY/dtheta,
        # the law gradient exact per node (:None) or interpolated linearly in Hbar on create_interpolation's knots (:Linear,
        # the target's default) -- the same two branches as dDiffusivity/dtheta
        nH, nS = _hyb_exps(law, ph)
        spatial = gamma_up_no_A(ph) * _pow(Hbar, nH + 1.0) * _pow(gS, nS - 1.0) * gSdV
        kind, nhalf = law.interp()
        if kind == "linear" and Hbar.max() > 0.0:
            g = law_grad_theta_linear(law, ph, Hbar, gS, theta, nhalf)
        else:
            g = law_grad_theta(law, ph, Hbar, gS, theta)
        return -np.tensordot(g, spatial, axes=([1, 2], [0, 1]))
    spatial = gamma_up_no_A(ph) * _pow(Hbar, ph.n + 1.0) * _pow(gS, ph.n - 1.0) * gSdV
    if law.kind == LAW_CONST_A:
        return -np.array([np.sum(spatial)])
    g = law_grad_theta(law, ph, Hbar, gS, theta)
    if law.kind == LAW_NN_A_SCALAR:
        return -g.reshape(-1) * np.sum(spatial)
    return -np.tensordot(g, spatial, axes=([1, 2], [0, 1]))


@dataclass
class LossVSpec:
    """LossV(loss=L2Sum, component, scale_loss)  (Losses.jl:66-81)."""

    component: str = "xy"
    scale_loss: bool = True
    log_eps: Optional[float] = None  # None: L2Sum; eps: LogSum(eps) (component "abs" only, Losses.jl:214)


def _lossV_scale(spec: LossVSpec, Vx_ref, Vy_ref, mask):
    if not spec.scale_loss:
        return 1.0
    return 1.0 / math.sqrt(np.mean(Vx_ref[mask] ** 2 + Vy_ref[mask] ** 2))


def loss_V(spec: LossVSpec, H, B, dx, dy, ph, law, Vabs_ref, Vx_ref, Vy_ref, normalization, theta=None):
    """loss(::LossV, ...) without the Dt factor (Losses.jl:293-337)."""
    Vx, Vy, V = V_from_H(H, B, dx, dy, ph, law, theta)
    mask = Vabs_ref > 0.0
    if spec.component == "xy":
        assert spec.log_eps is None
        l = l2sum_loss(Vx, Vx_ref, mask, normalization) + l2sum_loss(Vy, Vy_ref, mask, normalization)
    elif spec.log_eps is not None:
        l = logsum_loss(V, Vabs_ref, mask, normalization, spec.log_eps)
    else:
        l = l2sum_loss(V, Vabs_ref, mask, normalization)
    return l * _lossV_scale(spec, Vx_ref, Vy_ref, mask)


def backward_loss_V(spec: LossVSpec, H, B, dx, dy, ph, law, Vabs_ref, Vx_ref, Vy_ref, normalization, theta=None):
    """backward_loss(::LossV, ...) without the Dt factor (Losses.jl:338-390): (dL/dH, dL/dtheta)."""
    Vx, Vy, V = V_from_H(H, B, dx, dy, ph, law, theta)
    mask = Vabs_ref > 0.0
    if spec.component == "xy":
        dVx = l2sum_backward(Vx, Vx_ref, mask, normalization)
        dVy = l2sum_backward(Vy, Vy_ref, mask, normalization)
    else:
        dV = (l2sum_backward(V, Vabs_ref, mask, normalization) if spec.log_eps is None
              else logsum_backward(V, Vabs_ref, mask, normalization, spec.log_eps))
        with np.errstate(divide="ignore", invalid="ignore"):
            dVx = np.where(mask, dV * (Vx - Vx_ref) / (V - Vabs_ref), 0.0)
            dVy = np.where(mask, dV * (Vy - Vy_ref) / (V - Vabs_ref), 0.0)
    sc = _lossV_scale(spec, Vx_ref, Vy_ref, mask)
    dVx, dVy = dVx * sc, dVy * sc
    return (vjp_surface_V_H(dVx, dVy, H, B, dx, dy, ph, law, theta),
            vjp_surface_V_theta(dVx, dVy, H, B, dx, dy, ph, law, theta))


def loss_and_grad_HV(gl: Glacier, law: Law, cfg: SimConfig, H_ref, tH_ref, V_ref=None, tV_ref=(), vspec=None,
                     loss_kind="H", scaling=1.0, theta=None):
    """SIA2D_grad_batch! (gradient.jl:45-275) with LossH, LossV or LossHV (Losses.jl:395-440).
    V_ref = list of (Vabs, Vx, Vy) at times tV_ref.  Weights: LossH wH = dtH; LossV wV = dtV;
    LossHV wH = dtH^2, wV = scaling*dtV^2 (the reference multiplies by Dt twice, Losses.jl:407,424-431)."""
    snaps, st, inc = forward(gl, law, cfg, theta)
    t = list(cfg.tstops)
    k = len(t)
    N = float(gl.B.size)
    dtH = loss_weights(t, tH_ref) if loss_kind in ("H", "HV") else [0.0] * k
    dtV = loss_weights(t, tV_ref) if loss_kind in ("V", "HV") else [0.0] * k
    wH = [d * d if loss_kind == "HV" else d for d in dtH]
    wV = [scaling * d * d if loss_kind == "HV" else d for d in dtV]
    tH, tV = list(tH_ref), list(tV_ref)
    P = 1 if law.kind == LAW_CONST_A else law.mlp.n_params
    dLdtheta = np.zeros(P)
    lam = [np.zeros_like(gl.B) for _ in range(k)]
    loss_tot = 0.0
    for j in reversed(range(k)):
        tj = t[j]
        if cfg.mb is not None and tj in cfg.mb_times:
            H_pre = snaps[j] - inc[tj]
            lam[j] = lam[j] + vjp_mb(cfg.mb, lam[j], H_pre, gl.B)
        dl = np.zeros_like(gl.B)
        dth = np.zeros(P)
        if wH[j] != 0.0:
            Hr = H_ref[tH.index(tj)]
            mask = is_in_glacier(Hr, cfg.loss_distance)
            dl = dl + hloss_backward(snaps[j], Hr, mask, N, cfg.h_log_eps) * wH[j]
            loss_tot += hloss(snaps[j], Hr, mask, N, cfg.h_log_eps) * wH[j]
        if wV[j] != 0.0:
            Va, Vxr, Vyr = V_ref[tV.index(tj)]
            gH, gth = backward_loss_V(vspec, snaps[j], gl.B, gl.dx, gl.dy, gl.phys, law, Va, Vxr, Vyr, N, theta)
            dl = dl + gH * wV[j]
            dth = dth + gth * wV[j]
            loss_tot += loss_V(vspec, snaps[j], gl.B, gl.dx, gl.dy, gl.phys, law, Va, Vxr, Vyr, N, theta) * wV[j]
        g = vjp_H(lam[j], snaps[j], gl.B, gl.dx, gl.dy, gl.phys, law, theta)
        if j > 0:
            dt = t[j] - t[j - 1]
            lam[j - 1] = lam[j] + dt * g + dl
            dLdtheta += dt * vjp_theta(lam[j - 1], snaps[j], gl.B, gl.dx, gl.dy, gl.phys, law, theta)
        dLdtheta += dth  # gradient.jl:252
    return loss_tot, dLdtheta, lam[0]
