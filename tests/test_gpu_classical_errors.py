"""GPU: classical per-glacier inversions (PerGlacierModel: GlacierWideInv / GriddedInv with
LawA(params), Laws.jl:402-460; aggregate rule Model.jl:208-224) and the C ABI's error behaviour."""
import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu


def _glaciers(odinn, shapes, As):
    out = []
    for k, ((nx, ny), A) in enumerate(zip(shapes, As)):
        H0, B = O.synthetic_alpine(nx, ny, hmax=160.0, slope=0.1)
        out.append(odinn.Glacier2D(f"SYN-{k}", H0, B, 50.0, 50.0, A=A))
    return out


def _params(odinn, k=7, step=1.0 / 96.0, epochs=60):
    p = odinn.Parameters(simulation=odinn.SimulationParameters(tspan=(2010.0, 2010.0 + (k - 1) * step)),
                         solver=odinn.SolverParameters(reltol=1e-10, step=step),
                         hyper=odinn.Hyperparameters(optimizer=odinn.LBFGS(), epochs=epochs))
    p.UDE.grad = odinn.DiscreteAdjoint()  # compared with the oracle's discrete reverse loop
    return p


def test_glacier_wide_inversion_gradient_and_recovery(gpu):
    p = _params(gpu)
    A_true = [4e-17, 1.5e-17]
    shapes = [(48, 40), (64, 48)]
    gl = _glaciers(gpu, shapes, A_true)
    res = gpu.run_b(gpu.Prediction(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.ConstantA())), gl, p))
    for g, r in zip(gl, res):
        g.thicknessData = gpu.ThicknessData(r.t, r.H)
        g.A = 2.0e-17  # first guess
    reg = gpu.GlacierWideInv(p, gl, "A")
    inv = gpu.Inversion(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.LawA(p, scalar=True)), regressors={"A": reg}), gl, p)
    # gradient vs the oracle: dL/dtheta_g = G_g * dA/dtheta_g, one slot per glacier
    th0 = reg.theta.copy()
    dth = np.zeros_like(th0)
    L0 = gpu.SIA2D_grad_b(dth, th0, inv)
    ph = O.Phys()
    lo, hi = ph.minA, ph.maxA
    Lo, go = 0.0, np.zeros(2)
    for k, g in enumerate(gl):
        A = lo + (hi - lo) * (np.tanh(th0[k]) + 1) / 2
        glo = O.Glacier(g.H0, g.B, 50.0, 50.0, ph)
        cfg = O.SimConfig(tstops=list(res[k].t), reltol=1e-10)
        L1, G1, _ = O.loss_and_grad(glo, O.Law(kind=O.LAW_CONST_A, A=A), cfg, res[k].H, res[k].t)
        Lo += L1
        go[k] = G1[0] * (hi - lo) / 2 * (1 - np.tanh(th0[k]) ** 2)
    assert abs(L0 - Lo) <= 1e-6 * Lo
    assert np.allclose(dth, go, rtol=1e-5)
    st = gpu.run_b(inv)
    assert min(st.losses) < 1e-6 * st.losses[0]
    A_fit = lo + (hi - lo) * (np.tanh(st.θ) + 1) / 2
    assert np.allclose(A_fit, A_true, rtol=1e-3)


def test_gridded_inversion_gradient_field(gpu):
    """dL/dA on the dual grid == sum_j dt_j * spat * D_adjoint (adjoint.jl:235-250 with the sparse
    tensor of target_utils.jl:163-173)."""
    p = _params(gpu, k=4)
    gl = _glaciers(gpu, [(48, 40)], [3e-17])
    res = gpu.run_b(gpu.Prediction(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.ConstantA())), gl, p))
    gl[0].thicknessData = gpu.ThicknessData(res[0].t, res[0].H)
    gl[0].A = 2e-17
    reg = gpu.GriddedInv(p, gl, "A")
    inv = gpu.Inversion(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.LawA(p, scalar=False)), regressors={"A": reg}), gl, p)
    rng = np.random.default_rng(0)
    th0 = reg.theta + 0.05 * rng.standard_normal(reg.n_params)
    dth = np.zeros_like(th0)
    gpu.SIA2D_grad_b(dth, th0, inv)
    # oracle: reverse loop with a gridded A, accumulating spat * Da per node
    ph = O.Phys()
    lo, hi = ph.minA, ph.maxA
    g = gl[0]
    Af = (lo + (hi - lo) * (np.tanh(th0) + 1) / 2).reshape((g.nx - 1, g.ny - 1), order="F")
    law = O.Law(kind=O.LAW_CONST_A, A=Af)
    glo = O.Glacier(g.H0, g.B, 50.0, 50.0, ph)
    ts = list(res[0].t)
    cfg = O.SimConfig(tstops=ts, reltol=1e-10)
    snaps, _, _ = O.forward(glo, law, cfg)
    w = O.loss_weights(ts, ts)
    N = float(g.B.size)
    lam = np.zeros_like(g.B)
    field = np.zeros_like(Af)
    for j in reversed(range(1, len(ts))):
        mask = O.is_in_glacier(res[0].H[j], 3)
        dl = O.l2sum_backward(snaps[j], res[0].H[j], mask, N) * w[j]
        dt = ts[j] - ts[j - 1]
        lam_new = lam + dt * O.vjp_H(lam, snaps[j], g.B, 50.0, 50.0, ph, law) + dl
        Hc, S, gSx, gSy, gS, Hbar, ex, ey, exc, eyc = O._forward_intermediates(snaps[j], g.B, 50.0, 50.0, ph)
        _, _, Da = O._D_adjoint(lam_new, 50.0, 50.0, exc, eyc)
        field += dt * O.dD_dlaw(law, ph, Hbar, gS) * Da
        lam = lam_new
    dA = (hi - lo) / 2 * (1 - np.tanh(th0) ** 2)
    ref = field.ravel(order="F") * dA
    assert rel_l2(dth, ref) < 1e-5


def test_abi_error_behaviour(gpu):
    L = gpu._lib
    b = gpu.GlacierBatch([(32, 24)], [50.0])
    H0, B = O.synthetic_alpine(32, 24)
    b.set_fields(0, H0, B)
    with pytest.raises(gpu.OdinnError, match="strictly increasing"):
        b.solve([0.0, 1.0, 0.5])
    with pytest.raises(gpu.OdinnError, match="at least 2"):
        b.solve([0.0])
    import ctypes
    Hf = np.asfortranarray(H0)
    out = np.empty_like(Hf)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = L.lib().odinn_sia2d_dhdt(b._h, 3, Hf.ctypes.data_as(dp), 0.0, out.ctypes.data_as(dp))
    assert rc == 1 and b"out of range" in L.lib().odinn_last_error()  # ODINN_ERR_ARG, message kept
    rc = L.lib().odinn_sia2d_dhdt(b._h, 0, None, 0.0, out.ctypes.data_as(dp))
    assert rc == 1 and b"null" in L.lib().odinn_last_error()
    with pytest.raises(gpu.OdinnError, match="no solve"):
        b.snapshot(0, 0)
    with pytest.raises(gpu.OdinnError, match="no trainable law"):
        b.set_theta(np.zeros(3))
    with pytest.raises(gpu.OdinnError, match="reference thickness"):
        b.loss_grad([0.0, 0.1])
    mlp = gpu.MLPSpec([1, 3, 1], [1, 2], None, gpu.POST_AFFINE, 8e-21, 8e-17)
    with pytest.raises(gpu.OdinnError, match="architecture needs"):
        b.set_law(gpu.LAW_NN_A_SCALAR, mlp, np.zeros(5))
    with pytest.raises(gpu.OdinnError, match="expects 2 MLP inputs"):
        b.set_law(gpu.LAW_NN_Y, mlp, np.zeros(mlp.n_params))
    b.set_mass_balance(0, np.zeros((32, 24)))
    with pytest.raises(gpu.OdinnError, match="not inside"):
        b.solve([0.0, 0.1], mb_times=[0.2])  # (a time inside tspan that is not a tstop is a stop of the integrator only)
    with pytest.raises(gpu.OdinnError, match="strictly increasing"):
        b.solve([0.0, 0.1], mb_times=[0.05, 0.05])
    with pytest.raises(gpu.OdinnError, match="same tspan"):
        b.set_glacier_stops(0, [0.0, 0.05])
        b.solve([0.0, 0.1])
    b.set_glacier_stops(0, None)
    with pytest.raises(gpu.OdinnError, match="maxiters"):
        b.solve([0.0, 50.0], maxiters=3, dt0=1e-9)
    # a tolerance no step size can meet: the solve is given up with ODINN_ERR_DTMIN (error 8) after 256 attempts in a row that did not
    # advance t (OrdinaryDiffEq: ReturnCode.DtLessThanMin, at the first dt <= eps(t)) -- under every forward schedule, and in the reverse solve
    for sched, scheme in ((dict(), 0), (dict(), 1), (dict(step_sc=0), 0), (dict(step_sc=1), 0), (dict(fused_tiles=1), 0)):
        b.set_schedule(**sched)
        with pytest.raises(gpu.OdinnError, match=r"error 8: the solve is stuck"):
            b.solve([2010.0, 2010.5], reltol=1e-30, abstol=1e-300, scheme=scheme)
    b.set_schedule()
    ts = [2010.0, 2010.25, 2010.5]
    b.set_reference(0, ts, [H0 * (1.0 - 0.01 * j) for j in range(3)], 3)
    for sched in (dict(), dict(adj_sc=1), dict(adj_sc=0), dict(adj_fused=0)):
        b.set_schedule(**sched)
        with pytest.raises(gpu.OdinnError, match=r"error 8: the reverse solve is stuck"):
            b.loss_grad_continuous(ts, reltol=1e-8, adj_reltol=1e-30, adj_abstol=1e-300, n_quadrature=6)
    b.set_schedule()
    L0, g0 = b.loss_grad_continuous(ts, reltol=1e-8, n_quadrature=6)  # (the batch is usable afterwards)
    assert np.isfinite(L0)
    with pytest.raises(ValueError):
        b.set_fields(0, H0[:-1], B)
    with pytest.raises(gpu.OdinnError):
        gpu.GlacierBatch([(2, 2)], [50.0])
    b.close()


def test_ice_free_glacier_and_empty_reference(gpu):
    """Degenerate inputs: a glacier with no ice at all integrates trivially; zero loss weights give
    a zero gradient without error."""
    b = gpu.GlacierBatch([(40, 30), (40, 30)], [50.0, 50.0], T=[-5.0, -5.0])
    H0, B = O.synthetic_alpine(40, 30)
    b.set_fields(0, np.zeros_like(H0), B)
    b.set_fields(1, H0, B)
    ts = [0.0, 0.25, 0.5]
    st = b.solve(ts, reltol=1e-8)
    assert np.all(b.snapshot(0, 2) == 0.0) and st[0].t_final == 0.5
    ph = O.Phys()
    mlp = O.default_nn(1, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
    b.set_law(gpu.LAW_NN_A_SCALAR, gpu.MLPSpec(mlp.widths, mlp.acts, None, O.POST_AFFINE, ph.minA, ph.maxA),
              mlp.init_theta(np.random.default_rng(1)))
    for k in range(2):
        b.set_reference(k, [0.0], [b.snapshot(k, 0)], 3)  # only the first data time -> weight 0
    L, g = b.loss_grad(ts, reltol=1e-8)
    assert L == 0.0 and np.all(g == 0.0)
    b.close()


def test_scalar_A_after_a_gridded_A_is_not_ignored(gpu):
    """Once any glacier of a batch carries a gridded A all kernels read the A field; a later odinn_set_A must replace
    that glacier's slice of the field (it used to be ignored silently)."""
    H0, B = O.synthetic_alpine(48, 40, hmax=160.0, slope=0.1)
    b = gpu.GlacierBatch([(48, 40), (48, 40)], [50.0, 50.0], A=[2e-17, 2e-17])
    for k in range(2):
        b.set_fields(k, H0, B)
    b.set_A_field(0, np.full((47, 39), 5e-17, order="F"))
    b.set_A(1, 3e-17)
    b.set_A(0, 4e-17)
    ph = O.Phys()
    for k, A in ((0, 4e-17), (1, 3e-17)):
        ref = O.sia2d_rhs(H0, B, 50.0, 50.0, ph, O.Law(kind=O.LAW_CONST_A, A=A))
        assert rel_l2(b.dhdt(k, H0), ref) < 1e-12
    b.close()
