"""GPU: the CFL-limited explicit Euler scheme (odinn_solver_opts.scheme = ODINN_SCHEME_EULER_CFL,
the north star's "CFL" mode; SURVEY 8(d) 24 B per cell-step) against its restatement in the oracle
(solve_euler_cfl -- own definition, the reference has no such scheme) and against the adaptive
RDPK3Sp35 solution it must converge to."""
import numpy as np
import pytest

from conftest import rel_l2
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,cfl", [((96, 80), 0.25), ((65, 37), 0.5), ((130, 50), 0.1)])
def test_euler_cfl_matches_oracle(gpu, shape, cfl):
    nx, ny = shape
    H0, B = O.synthetic_valley(nx, ny, 50.0)
    ph = O.Phys()
    A = 2.21e-18 * 10
    law = O.Law(kind=O.LAW_CONST_A, A=A)
    ts = [2010.0, 2010.0 + 1 / 12.0, 2010.0 + 2 / 12.0]
    snaps, nsteps = O.solve_euler_cfl(O.Glacier(H0, B, 50.0, 50.0, ph), law, ts, cfl=cfl)
    b = gpu.GlacierBatch([(nx, ny)], [50.0], A=[A])
    b.set_fields(0, H0, B)
    st = b.solve(ts, scheme=gpu._lib.SCHEME_EULER_CFL, cfl=cfl)
    assert st[0].naccept == nsteps and st[0].nreject == 0 and st[0].t_final == ts[-1]
    for j in range(3):
        assert rel_l2(b.snapshot(0, j), snaps[j]) < 1e-11
    # first-order convergence towards the adaptive high-order solution
    b.solve(ts, reltol=1e-10)
    ref = b.snapshot(0, 2)
    assert rel_l2(snaps[2], ref) < 5e-3
    b.close()


def test_euler_cfl_with_mass_balance_and_batch(gpu):
    from test_gpu_parity import _mb

    shapes = [(96, 80), (64, 48)]
    ph = O.Phys()
    ts = [2010.0 + k / 12.0 for k in range(4)]
    b = gpu.GlacierBatch(shapes, [50.0, 50.0], A=[3e-17, 1e-17])
    refs = []
    for k, (nx, ny) in enumerate(shapes):
        H0, B = O.synthetic_valley(nx, ny, 50.0)
        b.set_fields(k, H0, B)
        mb = _mb(H0, B)
        b.set_mass_balance(k, mb.mb0, mb.dmb_dS, mb.S_ref, mb.mb_max)
        cb = lambda u, t, mb=mb, B=B: O.mb_apply(mb, u, B)[0]
        snaps, n = O.solve_euler_cfl(O.Glacier(H0, B, 50.0, 50.0, ph), O.Law(kind=O.LAW_CONST_A, A=[3e-17, 1e-17][k]), ts,
                                     cfl=0.3, callback=cb, callback_times=ts[1:])
        refs.append((snaps, n))
    st = b.solve(ts, mb_times=ts[1:], scheme=gpu._lib.SCHEME_EULER_CFL, cfl=0.3)
    for k in range(2):
        assert st[k].naccept == refs[k][1]  # independent step sequences inside one batch
        assert rel_l2(b.snapshot(k, 3), refs[k][0][3]) < 1e-11
    b.close()


def test_euler_cfl_rejects_bad_cfl(gpu):
    H0, B = O.synthetic_valley(32, 24, 50.0)
    b = gpu.GlacierBatch([(32, 24)], [50.0])
    b.set_fields(0, H0, B)
    with pytest.raises(Exception):
        b.solve([0.0, 0.1], scheme=gpu._lib.SCHEME_EULER_CFL, cfl=1.5)
    b.close()
