"""Generates the committed golden vectors tests/golden/*.npz from the CPU oracle.

The reference (Julia) cannot be executed in the build image, so these are ORACLE outputs,
pinned here so that (a) the oracle cannot drift silently and (b) the GPU path is checked
against fixed numbers as well as against the live oracle.  Inputs follow SURVEY 8(c)(5):
(H, B, lambda) -> (dH, J_H^T lambda, J_theta^T lambda) on <= 64x48 grids, seed 1234.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sia2d_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def case_inputs(name):
    rng = np.random.default_rng(1234)
    ph = O.Phys()
    if name == "valley_constA":
        H, B = O.synthetic_valley(64, 48, 50.0)
        law = O.Law(kind=O.LAW_CONST_A, A=2.21e-18)
        dx = 50.0
    elif name == "icecap_sliding":
        H, B = O.synthetic_icecap(48, 40, 100.0)
        H = H * 0.4
        ph = O.Phys(C=7e-8, p=3.0, q=1.0)
        law = O.Law(kind=O.LAW_CONST_A, A=2.21e-18)
        dx = 100.0
    elif name == "valley_nnA":
        H, B = O.synthetic_valley(64, 48, 50.0)
        mlp = O.default_nn(1, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
        th = mlp.init_theta(np.random.default_rng(7)) + 0.05 * np.random.default_rng(8).standard_normal(mlp.n_params)
        law = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th, T=-3.0)
        dx = 50.0
    elif name == "rough_random":
        H = np.asfortranarray(np.abs(rng.standard_normal((33, 29))) * 40.0 - 8.0)
        B = np.asfortranarray(1000.0 + 25.0 * rng.standard_normal((33, 29)))
        law = O.Law(kind=O.LAW_CONST_A, A=1e-17)
        dx = 80.0
    elif name == "valley_nnY":
        H, B = O.synthetic_valley(48, 40, 50.0)
        mlp = O.MLP([2, 3, 10, 3, 1], [1, 1, 1, 2], [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
        th = mlp.init_theta(np.random.default_rng(9)) + 0.05 * np.random.default_rng(10).standard_normal(mlp.n_params)
        law = O.Law(kind=O.LAW_NN_Y, mlp=mlp, theta=th, T=-5.0)
        dx = 50.0
    else:
        raise KeyError(name)
    lam = np.asfortranarray(np.random.default_rng(4321).standard_normal(H.shape))
    return H, B, lam, dx, ph, law


CASES = ["valley_constA", "icecap_sliding", "valley_nnA", "rough_random", "valley_nnY"]


def compute(name):
    H, B, lam, dx, ph, law = case_inputs(name)
    return dict(
        H=H, B=B, lam=lam, dx=dx,
        dH=O.sia2d_rhs(H, B, dx, dx, ph, law),
        vjp_H=O.vjp_H(lam, H, B, dx, dx, ph, law),
        vjp_theta=O.vjp_theta(lam, H, B, dx, dx, ph, law),
        theta=np.zeros(0) if law.theta is None else law.theta,
    )


def compute_velocity(name):
    """Surface-velocity seam of the A-type cases (not stored in the .npz files; used by the reference-dump comparison):
    (Vx, Vy) = V_from_H(H) and VJP_lambda_dsurface_V/d{H, theta} with the cotangents (dVx, dVy) = (lam, lam reversed)."""
    H, B, lam, dx, ph, law = case_inputs(name)
    if law.kind not in (O.LAW_CONST_A, O.LAW_NN_A_SCALAR):
        return None
    dVx, dVy = lam, np.asfortranarray(lam[::-1, :])
    Vx, Vy, _ = O.V_from_H(H, B, dx, dx, ph, law)
    return dict(Vx=Vx, Vy=Vy, vjp_surfV_H=O.vjp_surface_V_H(dVx, dVy, H, B, dx, dx, ph, law),
                vjp_surfV_theta=O.vjp_surface_V_theta(dVx, dVy, H, B, dx, dx, ph, law))


def solve_case():
    """Short adaptive solve + discrete adjoint on a 48x40 valley (snapshots, loss, gradient)."""
    ph = O.Phys()
    H0, B = O.synthetic_valley(48, 40, 50.0)
    ts = [2010.0 + j / 48.0 for j in range(5)]
    mlp = O.default_nn(1, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    th_true = mlp.init_theta(np.random.default_rng(42))
    th0 = mlp.init_theta(np.random.default_rng(1234))
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    cfg = O.SimConfig(tstops=ts, reltol=1e-8)
    ref, st, _ = O.forward(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th_true, T=-2.0), cfg)
    L, g, lam0 = O.loss_and_grad(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th0, T=-2.0), cfg, ref, ts)
    return dict(H0=H0, B=B, ts=np.array(ts), th_true=th_true, th0=th0, ref=np.stack(ref), loss=L, grad=g, lam0=lam0,
                naccept=st.naccept)


SOLVE_NQ = 24  # Gauss-Legendre nodes of the continuous-adjoint part of the whole-path dump
PIECES_NHALF = 75


def solve_case_continuous():
    """The same case through the continuous adjoint (ContinuousAdjoint(n_quadrature = SOLVE_NQ), reltol = abstol = 1e-8)."""
    c = solve_case()
    ph = O.Phys()
    mlp = O.default_nn(1, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    gl = O.Glacier(c["H0"], c["B"], 50.0, 50.0, ph)
    cfg = O.SimConfig(tstops=list(c["ts"]), reltol=1e-8)
    L, g, lam0, st = O.loss_and_grad_continuous(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=c["th0"], T=-2.0), cfg,
                                                list(c["ref"]), list(c["ts"]), O.ContinuousAdjointCfg(n_quadrature=SOLVE_NQ))
    return dict(loss=L, grad=g, lam0=lam0, naccept=st.naccept, nreject=st.nreject)


def solve_case_forward():
    """Forward solve of the case at theta0 (what run!(Prediction) / simulate_iceflow_UDE! produce): snapshots and step counts."""
    c = solve_case()
    ph = O.Phys()
    mlp = O.default_nn(1, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    gl = O.Glacier(c["H0"], c["B"], 50.0, 50.0, ph)
    snaps, st, _ = O.forward(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=c["th0"], T=-2.0), O.SimConfig(tstops=list(c["ts"]), reltol=1e-8))
    return dict(snaps=np.stack(snaps), naccept=st.naccept, nreject=st.nreject)


def pieces_inputs():
    """Inputs of the out-of-tree pieces (see oracle/julia/export_inputs.py::export_mask_cases)."""
    H_a, B_a = O.synthetic_valley(64, 48, 50.0)
    rng = np.random.default_rng(1234)
    H_b = np.asfortranarray(np.abs(rng.standard_normal((33, 29))) * 40.0 + 1.0)
    H_b[:, :4] = 0.0       # an ice-free strip; elsewhere the ice touches the border of the domain
    H_b[10:14, 12:16] = 0.0  # a hole
    H_b[25, 20] = 0.0        # a single ice-free cell
    MB = np.asfortranarray(1.5 * np.random.default_rng(77).standard_normal(H_a.shape))
    Hmb = np.asfortranarray(np.where(H_a > 0, H_a * np.random.default_rng(78).random(H_a.shape) * 0.2, 0.0))  # thin ice: the clip acts
    return dict(H_a=H_a, H_b=H_b, MB=MB, Hmb=Hmb)


def pieces_outputs():
    p = pieces_inputs()
    Hbar = O.avg(np.maximum(p["H_a"], 0.0))
    mb = O.MassBalance(mb0=p["MB"], dmb_dS=0.0, S_ref=None, mb_max=np.inf)
    Hn, inc = O.mb_apply(mb, p["Hmb"], np.zeros_like(p["Hmb"]))
    return dict(mask_a=O.is_in_glacier(p["H_a"], 3).astype(np.float64), mask_b=O.is_in_glacier(p["H_b"], 3).astype(np.float64),
                knots=np.asarray(O.create_interpolation(Hbar, PIECES_NHALF)), H_after_mb=Hn, mb_applied=inc)


if __name__ == "__main__":
    for c in CASES:
        np.savez_compressed(os.path.join(HERE, f"rhs_{c}.npz"), **compute(c))
    np.savez_compressed(os.path.join(HERE, "solve_valley_nnA.npz"), **solve_case())
    print("golden vectors written to", HERE)
