"""-m "not gpu": the oracle's restatement of LossAvgV, the second time-aggregated loss
(src/losses/TimeAggregatedLosses.jl:115-258; its place in both adjoints: src/inverse/SIA2D/gradient.jl:170-215,274,
:369-449,538), against its definition and finite differences of the loss."""
import numpy as np
import pytest

from conftest import stats_err_arrays
from oracle import sia2d_oracle as O
from test_oracle_dhdt import _fd
from test_oracle_gradient import _case


def _avgv_data(gl, law, cfg, ts, i1, i2, component, scale=1.1):
    """One velocity sample: the time average of the true run over [ts[i1], ts[i2]), scaled -- something to fit."""
    snaps, _, _ = O.forward(gl, law, cfg)
    a = O.AvgVData(ts[i1], ts[i2], None, None, None, component, ts[1] - ts[0])
    tl, dt = O.avgv_times(a)
    assert len(tl) == i2 - i1
    vx = sum(O.V_from_H(snaps[i1 + i], gl.B, gl.dx, gl.dy, gl.phys, law)[0] * dt[i] for i in range(len(tl))) / sum(dt)
    vy = sum(O.V_from_H(snaps[i1 + i], gl.B, gl.dx, gl.dy, gl.phys, law)[1] * dt[i] for i in range(len(tl))) / sum(dt)
    a.Vx, a.Vy = scale * vx, scale * vy
    a.Vabs = np.sqrt(a.Vx ** 2 + a.Vy ** 2)
    return a


def test_avgv_terms_definition():
    ph, gl, mlp, th_true, th0, ts, cfg, ref = _case(1.0 / 480.0, 13)
    law = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th0, T=-2.0)
    a = _avgv_data(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th_true, T=-2.0), cfg, ts, 2, 12, "xy")
    snaps, _, _ = O.forward(gl, law, cfg)
    cfg.avgv, cfg.avgv_weight = a, 2.5
    l, dl, dth = O.avgv_loss_terms(snaps, ts, cfg, gl, law)
    assert sorted(dl) == list(range(2, 12))  # tLoss = t1:step:t2 without its last point
    tl, dt = O.avgv_times(a)
    T = sum(dt)
    V = [O.V_from_H(snaps[j], gl.B, gl.dx, gl.dy, ph, law) for j in range(2, 12)]
    avx = sum(v[0] * d / T for v, d in zip(V, dt))
    avy = sum(v[1] * d / T for v, d in zip(V, dt))
    m = a.Vabs > 0
    want = 2.5 * (((avx - a.Vx)[m] ** 2).sum() + ((avy - a.Vy)[m] ** 2).sum()) / gl.B.size
    assert np.isclose(l, want, rtol=1e-13)
    # dL/dH of one stop by finite differences of the term itself
    rng = np.random.default_rng(0)
    e = rng.standard_normal(gl.B.shape) * (snaps[5] > 0)
    eps = 1e-5
    sp, sm = list(snaps), list(snaps)
    sp[5] = snaps[5] + eps * e
    sm[5] = snaps[5] - eps * e
    fd = (O.avgv_loss_terms(sp, ts, cfg, gl, law)[0] - O.avgv_loss_terms(sm, ts, cfg, gl, law)[0]) / (2 * eps)
    assert np.isclose(fd, (dl[5] * e).sum(), rtol=1e-6)
    # the :abs component
    cfg.avgv = _avgv_data(gl, O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th_true, T=-2.0), cfg, ts, 2, 12, "abs")
    l2, dl2, _ = O.avgv_loss_terms(snaps, ts, cfg, gl, law)
    av = np.sqrt(avx ** 2 + avy ** 2)
    assert np.isclose(l2, 2.5 * ((av - cfg.avgv.Vabs)[m] ** 2).sum() / gl.B.size, rtol=1e-13)
    fd = (O.avgv_loss_terms(sp, ts, cfg, gl, law)[0] - O.avgv_loss_terms(sm, ts, cfg, gl, law)[0]) / (2 * eps)
    # the reference's :abs cotangent is dl/dV (Vx - Vx_ref) / (V - V_ref), not the chain rule's Vx / V
    # (TimeAggregatedLosses.jl:229-231, the same form as LossV's): it equals the derivative only where the reference is
    # parallel to the prediction -- which the scaled true average used here is
    assert np.isclose(fd, (dl2[5] * e).sum(), rtol=2e-2)
    with pytest.raises(ValueError):
        cfg.avgv = O.AvgVData(ts[2] + 1e-4, ts[12], a.Vabs, a.Vx, a.Vy, "xy", ts[1] - ts[0])
        O.avgv_loss_terms(snaps, ts, cfg, gl, law)
    cfg.avgv = None


@pytest.mark.parametrize("component", ["xy", "abs"])
def test_avgv_gradient_vs_finite_differences_both_adjoints(component):
    """LossH + LossAvgV and LossAvgV alone: dL/dtheta of both adjoints against central differences of the loss."""
    ph, gl, mlp, th_true, th0, ts, cfg, ref = _case(1.0 / 480.0, 13)
    law = lambda th: O.Law(kind=O.LAW_NN_A_SCALAR, mlp=mlp, theta=th, T=-2.0)
    a = _avgv_data(gl, law(th_true), cfg, ts, 0, 12, component, scale=1.0)
    idx = np.arange(0, th0.size, 6)
    for with_H in ((True, False) if component == "xy" else (False,)):
        cfg.avgv, cfg.avgv_weight = a, (3.0 if with_H else 1.0)
        Href, tH = (ref, ts) if with_H else ([], [])

        def loss_at(th):
            s, _, _ = O.forward(gl, law(th), cfg)
            return (O.loss_H(s, ts, Href, tH, 3) if with_H else 0.0) + O.avgv_loss_terms(s, ts, cfg, gl, law(th))[0]

        gn = _fd(loss_at, th0, idx)
        L, g, _ = O.loss_and_grad(gl, law(th0), cfg, Href, tH)
        assert np.isclose(L, loss_at(th0), rtol=1e-12)
        ratio, angle, relerr = stats_err_arrays(g[idx], gn[idx])
        assert abs(ratio) < 2e-2 and abs(angle) < 1e-6 and relerr < 2e-2, (with_H, ratio, angle, relerr)
        Lc, gc, _, _ = O.loss_and_grad_continuous(gl, law(th0), cfg, Href, tH, O.ContinuousAdjointCfg(n_quadrature=200))
        assert np.isclose(Lc, L, rtol=1e-12)
        ratio, angle, relerr = stats_err_arrays(gc[idx], gn[idx])
        assert abs(ratio) < 5e-3 and abs(angle) < 1e-6 and relerr < 5e-3, (with_H, ratio, angle, relerr)
    cfg.avgv = None
