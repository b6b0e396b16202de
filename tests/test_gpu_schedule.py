"""GPU: odinn_set_schedule -- every kernel-selection switch of the library is a field of odinn_schedule (the ODINN_* environment
variables remain as overrides).  One small gradient per non-default schedule, through the ABI field (no environment), against
the automatic schedule: the equivalent kernel forms agree to rounding (fixed dt) / to the integration error (adaptive)."""
import numpy as np
import pytest

from conftest import rel_l2
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu

T0 = 2010.0
SHAPES = [(96, 80), (64, 48)]


def _case(gpu, law="scalar"):
    ph = O.Phys()
    k = 4
    ts = [T0 + j / 96.0 for j in range(k)]
    b = gpu.GlacierBatch(SHAPES, [50.0, 50.0], T=[-2.0, -4.0])
    refs = []
    for g, (nx, ny) in enumerate(SHAPES):
        H0, B = O.synthetic_valley(nx, ny, 50.0)
        b.set_fields(g, H0, B)
        refs.append([H0 * (1.0 - 0.01 * j) for j in range(k)])
        b.set_reference(g, ts, refs[-1], 3)
    if law == "scalar":
        om = O.default_nn(1, light=False, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
        b.set_law(gpu.LAW_NN_A_SCALAR, gpu.MLPSpec(om.widths, om.acts, None, O.POST_AFFINE, ph.minA, ph.maxA),
                  om.init_theta(np.random.default_rng(1234)))
    elif law == "gridded":
        om = O.default_nn(1, light=False, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
        b.set_law(gpu.LAW_NN_A_GRIDDED, gpu.MLPSpec(om.widths, om.acts, None, O.POST_AFFINE, ph.minA, ph.maxA),
                  om.init_theta(np.random.default_rng(1234)))
        for g, (nx, ny) in enumerate(SHAPES):
            xx = np.linspace(-6.0, -1.0, nx - 1)[:, None] + np.zeros((1, ny - 1))
            b.set_T_field(g, np.asfortranarray(xx))
    elif law == "Y":
        om = O.MLP([2, 3, 10, 3, 1], [1, 1, 1, 2], [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
        b.set_law(gpu.LAW_NN_Y, gpu.MLPSpec(om.widths, om.acts, [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA),
                  om.init_theta(np.random.default_rng(9)))
    return b, ts


def _grads(b, ts, continuous):
    if continuous:
        return b.loss_grad_continuous(ts, reltol=1e-8, n_quadrature=10)
    return b.loss_grad(ts, fixed_dt=1.0 / 1920.0)


GENERIC = [dict(step_sc=0), dict(step_sc=1), dict(fused_tiles=1), dict(fused_tiles=2), dict(fused_tiles=3), dict(fused_tiles=4),
           dict(dhdt_strip=0), dict(vjph_strip=0), dict(vjph_strip=1), dict(vjpth_strip=0), dict(vjpth_strip=1),
           dict(snap_on_load=0), dict(adj_fused=0), dict(adj_skip=0), dict(adj_segs=0), dict(adj_rows=4), dict(adj_rows=7), dict(adj_rows=8),
           dict(adj_theta_fused=0)]


@pytest.mark.parametrize("sched", GENERIC, ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_every_schedule_field_selects_an_equivalent_kernel_form(gpu, monkeypatch, sched):
    monkeypatch.delenv("ODINN_SCHEDULE", raising=False)
    b, ts = _case(gpu)
    assert all(v == -1 for v in b.get_schedule().values())
    Ld, gd = _grads(b, ts, False)
    Lc, gc = _grads(b, ts, True)
    b.set_schedule(**sched)
    eff = b.get_schedule()
    assert all(eff[k] == v for k, v in sched.items()) and sum(v != -1 for v in eff.values()) == len(sched)
    Ld2, gd2 = _grads(b, ts, False)
    Lc2, gc2 = _grads(b, ts, True)
    assert abs(Ld2 - Ld) <= 1e-11 * abs(Ld) and rel_l2(gd2, gd) < 1e-10, (sched, rel_l2(gd2, gd))
    assert abs(Lc2 - Lc) <= 1e-7 * abs(Lc) and rel_l2(gc2, gc) < 1e-6, (sched, rel_l2(gc2, gc))
    b.set_schedule()  # back to automatic: bitwise the first results again (same kernels, deterministic reductions)
    Ld3, gd3 = _grads(b, ts, False)
    assert Ld3 == Ld and np.array_equal(gd3, gd)
    # the environment (ODINN_SCHEDULE="field=value,...") overrides the field
    key, val = next(iter(sched.items()))
    if key != "fused_tiles":
        b.set_schedule(**{key: val})
        monkeypatch.setenv("ODINN_SCHEDULE", "dhdt_strip=1, %s=%s" % (key, "7" if key == "adj_rows" and val == 4 else ("4" if key == "adj_rows" else str(1 - val))))
        assert b.get_schedule()[key] != val
    b.close()


@pytest.mark.parametrize("law,sched", [("gridded", dict(lawgrad_wave=0)), ("Y", dict(interp_batch=0)),
                                       ("Y", dict(interp_batch=0, interp_streams=1))],
                         ids=["lawgrad_wave=0", "interp_batch=0", "interp_streams=1"])
def test_law_specific_schedule_fields(gpu, monkeypatch, law, sched):
    monkeypatch.delenv("ODINN_SCHEDULE", raising=False)
    b, ts = _case(gpu, law)
    L0, g0 = _grads(b, ts, False)
    b.close()
    b, ts = _case(gpu, law)  # (the interpolation scratch is sized at first use: a fresh batch for the forced form)
    b.set_schedule(**sched)
    L1, g1 = _grads(b, ts, False)
    b.close()
    assert abs(L1 - L0) <= 1e-11 * abs(L0) and rel_l2(g1, g0) < 1e-9, rel_l2(g1, g0)


def test_schedule_argument_checks(gpu):
    b, ts = _case(gpu)
    with pytest.raises(gpu.OdinnError, match="adj_rows"):
        b.set_schedule(adj_rows=5)
    with pytest.raises(TypeError):
        b.set_schedule(no_such_field=1)
    b.close()
