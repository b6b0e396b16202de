"""GPU: the non-default kernel schedules against the ORACLE.

tests/test_gpu_schedule.py compares the schedules with each other; the driver's `pytest -m gpu` otherwise sees the automatic
schedule only.  Here a small oracle-parity core -- one RHS evaluation, both VJPs, a fixed-step solve, a discrete-adjoint and a
continuous-adjoint gradient -- runs under every schedule that changes which kernel, hence which order of arithmetic, produces
them (odinn_schedule fields through the ABI, no environment): per-stage vs fused steps, LDS tiles vs strip kernels, the fused
reverse step with 2 / 4 / 7 rows per thread vs five stage launches, the self-controlled step loops vs the controller launches,
and for the Y law the table vs the network in the stencil and the sort-free vs sorted `:Linear` contraction.  The oracle is
evaluated once per case (module cache); tolerances are those of tests/test_gpu_parity.py / test_gpu_continuous_adjoint.py."""
import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu

NX, NY, DX = 72, 60, 50.0
_ORACLE = {}


def _inputs(law):
    ph = O.Phys()
    H0, B = O.synthetic_valley(NX, NY, DX)
    ts = [2010.0 + j / 96.0 for j in range(4)]
    ref = [H0 * (1.0 - 0.01 * j) for j in range(len(ts))]
    rng = np.random.default_rng(1234)
    lam = np.asfortranarray(rng.standard_normal((NX, NY)))
    if law == "A":
        om = O.default_nn(1, light=False, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
        olaw = O.Law(kind=O.LAW_NN_A_SCALAR, mlp=om, theta=om.init_theta(np.random.default_rng(42)), T=-2.0)
    else:
        # (a valley symmetric in y holds pairs of dual nodes whose Hbar agree to the last bit or not depending on the order the four
        #  cells are added in: whether ONE or TWO nodes attain max Hbar moves the count of values "strictly inside (0, max)" that
        #  create_interpolation takes its quantiles from, hence every quantile knot -- a 1e-7 effect of the reference's definition,
        #  not of a kernel.  A 1e-3 roughness removes the ties.)
        H0 = np.asfortranarray(H0 * (1.0 + 1e-3 * np.random.default_rng(3).random(H0.shape)))
        ref = [H0 * (1.0 - 0.01 * j) for j in range(len(ts))]
        om = O.MLP([2, 3, 10, 3, 1], [1, 1, 1, 2], [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
        olaw = O.Law(kind=O.LAW_NN_Y, mlp=om, theta=om.init_theta(np.random.default_rng(9)), T=-5.0)
    return ph, H0, B, ts, ref, lam, om, olaw


def _oracle(law):
    if law in _ORACLE:
        return _ORACLE[law]
    ph, H0, B, ts, ref, lam, om, olaw = _inputs(law)
    gl = O.Glacier(H0, B, DX, DX, ph)
    out = {"rhs": O.sia2d_rhs(H0, B, DX, DX, ph, olaw), "vjpH": O.vjp_H(lam, H0, B, DX, DX, ph, olaw),
           "vjpT": O.vjp_theta(lam, H0, B, DX, DX, ph, olaw)}
    cfg_fixed = O.SimConfig(tstops=ts, fixed_dt=1.0 / 1920.0)
    out["solve"] = O.forward(gl, olaw, cfg_fixed)[0][-1]
    out["disc"] = O.loss_and_grad(gl, olaw, cfg_fixed, ref, ts)[:2]
    cfg = O.SimConfig(tstops=ts, reltol=1e-8)
    out["cont"] = O.loss_and_grad_continuous(gl, olaw, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=8))[:2]
    _ORACLE[law] = out
    return out


SCHEDULES_A = [dict(), dict(solve_scheme=1), dict(fused_tiles=1), dict(fused_tiles=2), dict(fused_tiles=3), dict(fused_tiles=4),
               dict(step_sc=0), dict(step_sc=1, fused_tiles=3), dict(snap_on_load=0, step_sc=0), dict(dhdt_strip=0),
               dict(vjph_strip=0, vjpth_strip=0), dict(vjph_strip=1, vjpth_strip=1), dict(adj_fused=0), dict(adj_fused=1, adj_rows=7),
               dict(adj_fused=1, adj_rows=4), dict(adj_fused=1, adj_rows=2), dict(adj_fused=1, adj_sc=0), dict(adj_fused=1, adj_sc=1, adj_rows=4),
               dict(adj_fused=1, adj_skip=0), dict(adj_fused=1, adj_segs=0), dict(adj_fused=1, adj_theta_fused=0)]
SCHEDULES_Y = [dict(), dict(law_table=0), dict(law_table=0, solve_scheme=1), dict(interp_async=0), dict(interp_async=2),
               dict(interp_batch=0), dict(adj_fused=0), dict(adj_sc=1), dict(adj_sc=0), dict(fused_tiles=2), dict(vjph_strip=0, vjpth_strip=0)]


def _run(gpu, law, sched):
    ph, H0, B, ts, ref, lam, om, olaw = _inputs(law)
    sched = dict(sched)
    scheme = sched.pop("solve_scheme", 0)
    b = gpu.GlacierBatch([(NX, NY)], [DX], T=[olaw.T])
    b.set_fields(0, H0, B)
    b.set_law(olaw.kind, gpu.MLPSpec(om.widths, om.acts, om.prescale, om.post_kind, om.post_lo, om.post_hi), olaw.theta)
    b.set_reference(0, ts, ref, 3)
    b.set_schedule(**sched)
    o = _oracle(law)
    tolH = 1e-11 if law == "A" else 1e-7  # (the Y law's alpha holds the reference's 1e-4 forward difference)
    assert rel_l2(b.dhdt(0, H0), o["rhs"]) < 1e-11
    assert rel_l2(b.vjp_H(0, lam, H0), o["vjpH"]) < tolH
    assert rel_l2(b.vjp_theta(0, lam, H0), o["vjpT"]) < 1e-9
    b.solve(ts, fixed_dt=1.0 / 1920.0, scheme=scheme)
    assert rel_l2(b.snapshot(0, len(ts) - 1), o["solve"]) < 1e-11
    L, g = b.loss_grad(ts, theta=olaw.theta, fixed_dt=1.0 / 1920.0, scheme=scheme)
    Lo, go = o["disc"]
    assert abs(L - Lo) <= 1e-9 * abs(Lo)
    st = stats_err_arrays(g, go)
    assert abs(st[0]) < (1e-8 if law == "A" else 1e-6) and st[2] < (1e-8 if law == "A" else 1e-6), ("discrete", sched, st)
    L, g = b.loss_grad_continuous(ts, theta=olaw.theta, reltol=1e-8, n_quadrature=8, scheme=scheme)
    Lo, go = o["cont"]
    assert abs(L - Lo) <= 1e-6 * abs(Lo)
    st = stats_err_arrays(g, go)
    assert abs(st[0]) < 2e-4 and st[2] < 2e-4, ("continuous", sched, st)
    b.close()


@pytest.mark.parametrize("sched", SCHEDULES_A, ids=lambda s: ",".join(f"{k}={v}" for k, v in s.items()) or "automatic")
def test_schedules_against_the_oracle_A_law(gpu, sched):
    _run(gpu, "A", sched)


@pytest.mark.parametrize("sched", SCHEDULES_Y, ids=lambda s: ",".join(f"{k}={v}" for k, v in s.items()) or "automatic")
def test_schedules_against_the_oracle_Y_law(gpu, sched):
    _run(gpu, "Y", sched)
