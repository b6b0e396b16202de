"""GPU: Tikhonov regulariser kernel (Regularization.jl:92-126, 330-382) against the oracle, and the
initial-condition inversion (theta.IC, InitialCondition_utils.jl; gradient.jl:262-271) with
InitialThicknessRegularization / RheologyRegularization composed by MultiLoss."""
import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,use_mask", [((9, 10), True), ((9, 10), False), ((130, 67), True), ((3, 3), False),
                                            ((64, 5), True)])
def test_tikhonov_kernel_matches_oracle(gpu, shape, use_mask):
    rng = np.random.default_rng(11)
    a = np.asfortranarray(rng.standard_normal(shape))
    mask = rng.standard_normal(shape) >= 0 if use_mask else np.ones(shape, bool)
    b = gpu.GlacierBatch([(8, 8)], [50.0])
    l, g = b.tikhonov(a, 1.2, 1.8, mask if use_mask else None)
    lo, go = O.tikhonov_loss(a, 1.2, 1.8, mask), O.tikhonov_backward(a, 1.2, 1.8, mask)
    assert abs(l - lo) <= 1e-13 * max(abs(lo), 1e-300)
    assert np.abs(g - go).max() <= 1e-13 * max(np.abs(go).max(), 1e-300)
    b.close()


def _setup(gpu, filt="identity", k=5, step=1.0 / 96.0):
    p = gpu.Parameters(simulation=gpu.SimulationParameters(tspan=(2010.0, 2010.0 + (k - 1) * step)),
                       solver=gpu.SolverParameters(reltol=1e-10, step=step),
                       hyper=gpu.Hyperparameters(optimizer=gpu.LBFGS(), epochs=25))
    p.UDE.initial_condition_filter = filt
    p.UDE.grad = gpu.DiscreteAdjoint()  # compared with the oracle's discrete reverse loop
    gl = []
    for kk, (nx, ny) in enumerate([(48, 40), (40, 32)]):
        H0, B = O.synthetic_alpine(nx, ny, hmax=160.0, slope=0.1)
        gl.append(gpu.Glacier2D(f"SYN-{kk}", H0, B, 50.0, 50.0, A=3e-17))
    res = gpu.run_b(gpu.Prediction(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.ConstantA())), gl, p))
    for g, r in zip(gl, res):
        g.thicknessData = gpu.ThicknessData(r.t, r.H)
    return p, gl


@pytest.mark.parametrize("filt", ["identity", "softplus", "Zang1980"])
def test_initial_condition_gradient_matches_oracle(gpu, filt):
    """dL/dtheta.IC = lam(t0) * dH0/dtheta.IC + lambda_reg * VJP_lap(2 lap H0) (gradient.jl:262-271,
    Regularization.jl:166-190), data term weighted by the MultiLoss weight."""
    p, gl = _setup(gpu, filt)
    wd, wr = 1.5, 2e-3
    p.UDE.empirical_loss_function = gpu.MultiLoss(
        losses=(gpu.LossH(), gpu.InitialThicknessRegularization(t0=2010.0)), lambdas=(wd, wr))
    ic = gpu.InitialCondition(p, gl)
    rng = np.random.default_rng(2)
    inv = gpu.Inversion(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.ConstantA()), regressors={"IC": ic}), gl, p)
    th = ic.theta * (1.0 + 0.05 * rng.standard_normal(ic.theta.size)) + 0.3 * rng.standard_normal(ic.theta.size)
    dth = np.zeros_like(th)
    L = gpu.SIA2D_grad_b(dth, th, inv)
    ts = inv.tstops()
    off = 0
    Lo = 0.0
    for g in gl:
        n = g.nx * g.ny
        x = th[off:off + n].reshape((g.nx, g.ny), order="F")
        H0 = O.evaluate_H0(x, g.mask, filt)
        glo = O.Glacier(H0, g.B, 50.0, 50.0, O.Phys())
        cfg = O.SimConfig(tstops=ts, reltol=1e-10)
        l, _, lam0 = O.loss_and_grad(glo, O.Law(kind=O.LAW_CONST_A, A=3e-17), cfg, g.thicknessData.H, ts)
        go = wd * lam0 * O.evaluate_dH0(x, g.mask, filt) + wr * O.tikhonov_backward(H0, 50.0, 50.0, np.ones_like(H0, bool))
        Lo += wd * l + wr * O.tikhonov_loss(H0, 50.0, 50.0, np.ones_like(H0, bool))
        assert rel_l2(dth[off:off + n].reshape((g.nx, g.ny), order="F"), go) < 1e-6
        off += n
    assert abs(L - Lo) <= 1e-7 * abs(Lo)


def test_initial_condition_inversion_reduces_loss(gpu):
    p, gl = _setup(gpu)
    ic = gpu.InitialCondition(p, gl)
    ic.theta = ic.theta * 0.9  # first guess: 10 % too thin
    inv = gpu.Inversion(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.ConstantA()), regressors={"IC": ic}), gl, p)
    st = gpu.run_b(inv)
    assert st.losses[-1] < 1e-2 * st.losses[0]
    # the recovered initial state is closer to the truth than the first guess
    n0 = gl[0].nx * gl[0].ny
    H_fit = st.θ[:n0].reshape((gl[0].nx, gl[0].ny), order="F")
    assert rel_l2(H_fit, gl[0].H0) < 0.5 * rel_l2(0.9 * gl[0].H0, gl[0].H0)


def test_rheology_regularization_gridded(gpu):
    """MultiLoss(LossH, RheologyRegularization) on a GriddedInv classical inversion: the regulariser's
    gradient through A = minA + (maxA-minA)(tanh θ+1)/2 (Regularization.jl:280-310)."""
    p, gl = _setup(gpu)
    wr = 1e36  # A ~ 1e-17 and dx = 50 m: (lap A)^2 ~ 1e-41 per node
    p.UDE.empirical_loss_function = gpu.MultiLoss(losses=(gpu.LossH(), gpu.RheologyRegularization()), lambdas=(1.0, wr))
    reg = gpu.GriddedInv(p, gl, "A")
    rng = np.random.default_rng(4)
    th = reg.theta + 0.2 * rng.standard_normal(reg.theta.size)
    inv = gpu.Inversion(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.LawA(p, scalar=False)), regressors={"A": reg}), gl, p)
    d1 = np.zeros_like(th)
    L1 = gpu.SIA2D_grad_b(d1, th, inv)
    p.UDE.empirical_loss_function = gpu.LossH()
    d0 = np.zeros_like(th)
    L0 = gpu.SIA2D_grad_b(d0, th, inv)
    lo, hi = p.physical.minA, p.physical.maxA
    off = 0
    Lr = 0.0
    for g in gl:
        n = (g.nx - 1) * (g.ny - 1)
        t = th[off:off + n]
        A = (lo + (hi - lo) * (np.tanh(t) + 1) / 2).reshape((g.nx - 1, g.ny - 1), order="F")
        m = np.ones_like(A, bool)
        Lr += wr * O.tikhonov_loss(A, 50.0, 50.0, m)
        gr = wr * O.tikhonov_backward(A, 50.0, 50.0, m).ravel(order="F") * (hi - lo) * (1 - np.tanh(t) ** 2) / 2
        assert np.abs((d1 - d0)[off:off + n] - gr).max() <= 1e-9 * np.abs(gr).max()
        off += n
    assert abs((L1 - L0) - Lr) <= 1e-9 * Lr


@pytest.mark.parametrize("adjoint", ["discrete", "continuous"])
def test_initial_condition_gradient_vs_finite_differences(gpu, adjoint):
    """test_grad_finite_diff(...; train_initial_conditions = true) (runtests.jl:120-122,129-131; thresholds
    [5e-3, 5e-7, 5e-3] discrete, [5e-4, 1e-8, 5e-4] continuous): dL/dH0 = lambda(t0) against central finite
    differences of the GPU forward loss over a sample of ice-covered cells."""
    p, gl = _setup(gpu, "identity", k=5, step=1.0 / 480.0)
    p.UDE.grad = gpu.DiscreteAdjoint() if adjoint == "discrete" else gpu.ContinuousAdjoint(n_quadrature=40)
    gl = gl[:1]
    ic = gpu.InitialCondition(p, gl)
    inv = gpu.Inversion(gpu.Model(gpu.SIA2Dmodel(p, A=gpu.ConstantA()), regressors={"IC": ic}), gl, p)
    th = ic.theta * 0.97
    dth = np.zeros_like(th)
    gpu.SIA2D_grad_b(dth, th, inv)
    b = inv.batch()
    ts = inv.tstops()
    g = gl[0]

    def loss_at(x):
        b.set_fields(0, np.asfortranarray(x.reshape((g.nx, g.ny), order="F")), g.B)
        b.solve(ts, reltol=1e-10)
        return float(b.loss()[0])

    rng = np.random.default_rng(0)
    cells = rng.choice(np.flatnonzero(th > 20.0), 12, replace=False)
    gn = np.zeros(cells.size)
    for n_, c in enumerate(cells):
        e = np.zeros_like(th)
        e[c] = 1e-3
        gn[n_] = (loss_at(th + e) - loss_at(th - e)) / 2e-3
    ratio, angle, relerr = stats_err_arrays(dth[cells], gn)
    thr = (5e-3, 5e-5, 5e-3) if adjoint == "discrete" else (2e-3, 5e-5, 2e-3)
    assert abs(ratio) < thr[0] and abs(angle) < thr[1] and relerr < thr[2], (ratio, angle, relerr)
