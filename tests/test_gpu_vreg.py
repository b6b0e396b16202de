"""GPU: VelocityRegularization (src/losses/Regularization.jl:64-79,192-245; tested in the reference by
test/runtests.jl:213-221 with the ContinuousAdjoint) -- loss, dL/dH at the velocity-data stops and dL/dtheta (sum over the
stops in the DiscreteAdjoint, Gauss-Legendre quadrature on the interpolated state in the ContinuousAdjoint) against the
oracle's restatement: alone and next to LossH, a mass balance, a ragged batch with a gridded law, through the API."""
import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays
from oracle import sia2d_oracle as O
from test_gpu_parity import _inversion_case

pytestmark = pytest.mark.gpu


def _dummy_v(shape, n):
    z = [np.zeros(shape)] * n
    return z, z, z


@pytest.mark.parametrize("case", ["with_H", "alone", "with_H_mb", "sparse_times"])
def test_vreg_loss_and_gradients_match_oracle(gpu, case):
    nx, ny = 64, 48
    use_mb = case == "with_H_mb"
    ph, H0, B, ts, om, gm, th0, gl, mb, cfg, ref = _inversion_case(gpu, nx, ny, use_mb)
    tV = ts[1::2] if case == "sparse_times" else list(ts)
    cfg.vreg_times, cfg.vreg_distance, cfg.vreg_weight = tV, 3, 50.0
    alone = case == "alone"
    Href, tH = ([], []) if alone else (ref, ts)
    law0 = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=th0, T=-2.0)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], T=[-2.0])
    b.set_fields(0, H0, B)
    b.set_law(gpu.LAW_NN_A_SCALAR, gm, th0)
    if Href:
        b.set_reference(0, ts, ref, 3)
    if mb is not None:
        b.set_mass_balance(0, mb.mb0, mb.dmb_dS, mb.S_ref, mb.mb_max)
    b.set_velocity_reference(0, tV, *_dummy_v((nx, ny), len(tV)))  # only the dates matter
    b.set_velocity_regularization(50.0, 3)
    mbt = ts[1:] if use_mb else ()
    b.solve(ts, mb_times=mbt, reltol=1e-8)
    snaps, _, _ = O.forward(gl, law0, cfg)
    lo_fwd = (O.loss_H(snaps, ts, Href, tH, 3) if Href else 0.0) + O.vreg_loss_terms(snaps, ts, cfg, gl, law0)[0]
    assert abs(b.loss()[0] - lo_fwd) <= 1e-6 * abs(lo_fwd)
    # discrete adjoint (lambda(t0) is not compared on adaptive solves: see test_gpu_avgv.py)
    Lo, go, _ = O.loss_and_grad(gl, law0, cfg, Href, tH)
    Lg, gg = b.loss_grad(ts, theta=th0, mb_times=mbt, reltol=1e-8)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (case, ratio, angle, relerr)
    # continuous adjoint
    adj = O.ContinuousAdjointCfg(n_quadrature=16)
    Lo, go, lam0, _ = O.loss_and_grad_continuous(gl, law0, cfg, Href, tH, adj)
    Lg2, gg2 = b.loss_grad_continuous(ts, theta=th0, mb_times=mbt, reltol=1e-8, n_quadrature=16)
    assert abs(Lg2 - Lo) <= 1e-6 * abs(Lo)
    ratio, angle, relerr = stats_err_arrays(gg2, go)
    assert abs(ratio) < 1e-5 and abs(angle) < 1e-9 and relerr < 1e-5, (case, ratio, angle, relerr)
    assert rel_l2(b.lambda0(0), lam0) < 1e-3
    # the term really is in there
    b.set_velocity_regularization(0.0, 3)
    if Href:
        L2, g2 = b.loss_grad(ts, theta=th0, mb_times=mbt, reltol=1e-8)
        assert abs(L2 - Lg) > 1e-6 * abs(Lg) and rel_l2(g2, gg) > 1e-6
    else:
        with pytest.raises(gpu.OdinnError):
            b.loss_grad(ts, theta=th0, reltol=1e-8)
    b.close()


@pytest.mark.parametrize("distance", [0, 1, 3, 5])
def test_vreg_mask_across_wavefront_and_tile_boundaries(gpu, distance):
    """is_in_glacier(H, distance) inside k_vreg_prep: row ballots eroded by scalar shifts, the `distance` columns either side of
    a wavefront's 64 loaded by its first lanes, every eroded row shared by the rows of the wavefront within `distance` of
    it.  A 150 x 70 glacier (three column tiles, five row tiles) whose margin crosses the tile boundaries, with holes and
    ice up to the grid edge: loss and dL/dtheta of the term at two stops against the oracle."""
    ph = O.Phys()
    nx, ny = 150, 70
    H0, B = O.synthetic_icecap(nx, ny, 50.0)
    H0 = H0 * 0.3
    rng = np.random.default_rng(12)
    holes = rng.random((nx, ny)) < 0.004          # isolated ice-free cells inside the ice
    H0 = np.asfortranarray(np.where(holes, 0.0, H0))
    H0[:, :2] = np.maximum(H0[:, :2], 20.0)       # ice on the first rows: the window leaves the grid
    H0[60:70, :] = np.maximum(H0[60:70, :], 15.0)  # a band across the wavefront boundary at column 64
    ts = [2010.0, 2010.0 + 1.0 / 48.0]
    law = O.Law(kind=O.LAW_CONST_A, A=4e-17)
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    cfg = O.SimConfig(tstops=ts, reltol=1e-8, vreg_times=ts, vreg_distance=distance, vreg_weight=20.0)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], A=[4e-17])
    b.set_fields(0, H0, B)
    b.set_velocity_reference(0, ts, *_dummy_v((nx, ny), 2))
    b.set_velocity_regularization(20.0, distance)
    b.solve(ts, reltol=1e-8)
    snaps, _, _ = O.forward(gl, law, cfg)
    lo, _, gth = O.vreg_loss_terms(snaps, ts, cfg, gl, law)
    assert lo > 0.0 and abs(b.loss()[0] - lo) <= 1e-9 * abs(lo), distance
    Lg, gg = b.loss_grad(ts, reltol=1e-8)
    Lo, go, _ = O.loss_and_grad(gl, law, cfg, [], [])
    assert abs(Lg - Lo) <= 1e-9 * abs(Lo) and rel_l2(np.ravel(gg), np.ravel(go)) < 1e-6, distance
    b.close()


def test_vreg_ragged_batch_gridded_law(gpu):
    """Three ragged glaciers, hoisted gridded A = NN(T) (theta-part through the dual-grid accumulator), one glacier
    without velocity dates; batch result = sum of the per-glacier oracle results, both adjoints."""
    ph = O.Phys()
    from test_gpu_parity import _mlp_pair
    om, gm, th = _mlp_pair(gpu, [1, 3, 10, 3, 1], [1, 1, 1, 2], None, O.POST_AFFINE, ph.minA, ph.maxA)
    shapes = [(56, 40), (80, 48), (40, 33)]
    step = 1.0 / 48.0
    ts = [2010.0 + j * step for j in range(7)]
    tVs = [ts, ts[::2], []]
    rng = np.random.default_rng(4)
    b = gpu.GlacierBatch(shapes, [50.0] * 3)
    gls, laws, cfgs, refs = [], [], [], []
    for k, (nx, ny) in enumerate(shapes):
        H0, B = O.synthetic_alpine(nx, ny, hmax=150.0, slope=0.1)
        T = np.asfortranarray(-5.0 - 4.0 * rng.uniform(size=(nx - 1, ny - 1)))
        b.set_fields(k, H0, B)
        b.set_T_field(k, T)
        gl = O.Glacier(H0, B, 50.0, 50.0, ph)
        law = O.Law(kind=O.LAW_NN_A_GRIDDED, mlp=om, theta=th, T=T)
        cfg = O.SimConfig(tstops=ts, reltol=1e-8, vreg_times=tVs[k], vreg_distance=2, vreg_weight=30.0)
        ref, _, _ = O.forward(gl, law, cfg)
        ref = [r * (1.0 + 0.02 * j) for j, r in enumerate(ref)]
        b.set_reference(k, ts, ref, 3)
        if tVs[k]:
            b.set_velocity_reference(k, tVs[k], *_dummy_v((nx, ny), len(tVs[k])))
        gls.append(gl); laws.append(law); cfgs.append(cfg); refs.append(ref)
    b.set_law(gpu.LAW_NN_A_GRIDDED, gm, th)
    b.set_velocity_regularization(30.0, 2)
    Lo, go, Lc, gc = 0.0, 0.0, 0.0, 0.0
    for k in range(3):
        l, g, _ = O.loss_and_grad(gls[k], laws[k], cfgs[k], refs[k], ts)
        Lo, go = Lo + l, go + g
        l, g, _, _ = O.loss_and_grad_continuous(gls[k], laws[k], cfgs[k], refs[k], ts, O.ContinuousAdjointCfg(n_quadrature=8))
        Lc, gc = Lc + l, gc + g
    Lg, gg = b.loss_grad(ts, theta=th, reltol=1e-8)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo) and rel_l2(gg, go) < 1e-5
    Lg, gg = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
    assert abs(Lg - Lc) <= 1e-6 * abs(Lc) and rel_l2(gg, gc) < 1e-5
    b.close()


def test_multiloss_with_velocity_regularization_through_the_api(gpu):
    """The reference's documented example: MultiLoss((LossH(), VelocityRegularization()), (0.4, lambda)) through
    Inversion / SIA2D_grad_b (per-glacier classical law) against the oracle, DiscreteAdjoint."""
    k, step = 7, 1.0 / 96.0
    p = gpu.Parameters(simulation=gpu.SimulationParameters(tspan=(2010.0, 2010.0 + (k - 1) * step)),
                       solver=gpu.SolverParameters(reltol=1e-10, step=step),
                       hyper=gpu.Hyperparameters(optimizer=gpu.LBFGS(), epochs=3))
    p.UDE.grad = gpu.DiscreteAdjoint()
    p.UDE.empirical_loss_function = gpu.MultiLoss(losses=(gpu.LossH(), gpu.VelocityRegularization(distance=2)), lambdas=(0.4, 8.0))
    ts = [2010.0 + j * step for j in range(k)]
    ph = O.Phys()
    gl = []
    for kk, (nx, ny) in enumerate([(48, 40), (64, 48)]):
        H0, B = O.synthetic_alpine(nx, ny, hmax=160.0, slope=0.1)
        g = gpu.Glacier2D(f"SYN-{kk}", H0, B, 50.0, 50.0, A=3e-17)
        g.thicknessData = gpu.ThicknessData(ts, [H0 * (1.0 - 0.01 * j) for j in range(k)])
        tv = ts[kk::2]
        z = [np.zeros((nx, ny))] * len(tv)
        g.velocityData = gpu.VelocityData(t=tv, vabs=z, vx=z, vy=z)
        gl.append(g)
    reg = gpu.GlacierWideInv(p, gl, "A")
    inv = gpu.Inversion(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.LawA(p, scalar=True)), regressors={"A": reg}), gl, p)
    th = reg.theta.copy()
    dth = np.zeros_like(th)
    L = gpu.SIA2D_grad_b(dth, th, inv)
    lo, hi = ph.minA, ph.maxA
    Lo, go = 0.0, np.zeros(2)
    for kk, g in enumerate(gl):
        A = lo + (hi - lo) * (np.tanh(th[kk]) + 1) / 2
        cfg = O.SimConfig(tstops=ts, reltol=1e-10, vreg_times=list(g.velocityData.t), vreg_distance=2, vreg_weight=8.0 / 0.4)
        l1, g1, _ = O.loss_and_grad(O.Glacier(g.H0, g.B, 50.0, 50.0, ph), O.Law(kind=O.LAW_CONST_A, A=A), cfg, g.thicknessData.H, ts)
        Lo += 0.4 * l1
        go[kk] = 0.4 * g1[0] * (hi - lo) / 2 * (1 - np.tanh(th[kk]) ** 2)
    assert abs(L - Lo) <= 1e-6 * abs(Lo)
    assert np.allclose(dth, go, rtol=1e-5)
