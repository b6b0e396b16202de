"""GPU: rectangular cells (dx != dy).  Every stencil kernel carries the two spacings separately (the strip step kernel has a
dx == dy instantiation that drops the ratio): RHS, both VJPs, the fixed-dt solve under every forward kernel form, the
surface velocity and the discrete / continuous gradients against the oracle on a 60 m x 45 m grid."""
import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu
DX, DY = 60.0, 45.0


def _glacier(nx, ny):
    H0, B = O.synthetic_valley(nx, ny, 50.0)
    return np.asfortranarray(H0), np.asfortranarray(B)


@pytest.mark.parametrize("shape", [(96, 80), (65, 47)])
def test_rhs_and_vjps_on_rectangular_cells(gpu, shape):
    nx, ny = shape
    H0, B = _glacier(nx, ny)
    ph = O.Phys()
    law = O.Law(kind=O.LAW_CONST_A, A=3e-17)
    b = gpu.GlacierBatch([shape], [DX], [DY], A=[3e-17])
    b.set_fields(0, H0, B)
    lam = np.asfortranarray(np.random.default_rng(5).standard_normal(shape))
    assert rel_l2(b.dhdt(0, H0), O.sia2d_rhs(H0, B, DX, DY, ph, law)) < 1e-12
    assert rel_l2(b.vjp_H(0, lam, H0), O.vjp_H(lam, H0, B, DX, DY, ph, law)) < 1e-11
    assert rel_l2(b.vjp_theta(0, lam, H0), O.vjp_theta(lam, H0, B, DX, DY, ph, law)) < 1e-11
    Vx, Vy = b.surface_V(0, H0)
    Vxo, Vyo, _ = O.V_from_H(H0, B, DX, DY, ph, law)
    assert rel_l2(Vx, Vxo) < 1e-12 and rel_l2(Vy, Vyo) < 1e-12
    for method in (gpu._lib.VJP_CONTINUOUS,):
        b.set_vjp_method(method)
        assert rel_l2(b.vjp_H(0, lam, H0), O.vjp_H_continuous(lam, H0, B, DX, DY, ph, law)) < 1e-11
    b.close()


@pytest.mark.parametrize("sched", [dict(), dict(fused_tiles=1), dict(fused_tiles=2), dict(fused_tiles=3), dict(fused_tiles=4),
                                   dict(step_sc=0, fused_tiles=4)], ids=str)
@pytest.mark.parametrize("scheme", [1, 2])
def test_fixed_dt_solve_on_rectangular_cells_every_kernel_form(gpu, sched, scheme):
    if scheme == 1 and sched:
        pytest.skip("the per-stage schedule has one form")
    nx, ny = 128, 96
    H0, B = _glacier(nx, ny)
    ph = O.Phys()
    gl = O.Glacier(H0, B, DX, DY, ph)
    ts = [0.0, 0.01, 0.02]
    fo, _, _ = O.forward(gl, O.Law(kind=O.LAW_CONST_A, A=3e-17), O.SimConfig(tstops=ts, fixed_dt=1e-3))
    b = gpu.GlacierBatch([(nx, ny)], [DX], [DY], A=[3e-17])
    b.set_fields(0, H0, B)
    b.set_schedule(**sched)
    b.solve(ts, fixed_dt=1e-3, scheme=scheme)
    for j in (1, 2):
        assert rel_l2(b.snapshot(0, j), fo[j]) < 1e-12, (sched, scheme, j)
    b.close()


@pytest.mark.parametrize("adjoint", ["discrete", "continuous"])
def test_gradients_on_rectangular_cells(gpu, adjoint):
    nx, ny = 64, 48
    H0, B = _glacier(nx, ny)
    ph = O.Phys()
    gl = O.Glacier(H0, B, DX, DY, ph)
    om = O.default_nn(1, light=False, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    gm = gpu.MLPSpec(om.widths, om.acts, None, O.POST_AFFINE, ph.minA, ph.maxA)
    th = om.init_theta(np.random.default_rng(1234))
    ts = [2010.0 + j / 96.0 for j in range(5)]
    cfg = O.SimConfig(tstops=ts, reltol=1e-8)
    ref, _, _ = O.forward(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=om.init_theta(np.random.default_rng(42)), T=-2.0), cfg)
    law = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th, T=-2.0)
    b = gpu.GlacierBatch([(nx, ny)], [DX], [DY], T=[-2.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_A_SCALAR, gm, th)
    b.set_reference(0, ts, ref, 3)
    if adjoint == "discrete":
        Lo, go, _ = O.loss_and_grad(gl, law, cfg, ref, ts)
        Lg, gg = b.loss_grad(ts, theta=th, reltol=1e-8)
    else:
        Lo, go, _, _ = O.loss_and_grad_continuous(gl, law, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=12))
        Lg, gg = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=12)
    b.close()
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (ratio, angle, relerr)
