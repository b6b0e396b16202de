"""The oracle's C restatement (the cpu_baseline leg of bench.py) against the numpy oracle."""
import numpy as np
import pytest

from conftest import rel_l2
from oracle import sia2d_oracle as O

CO = pytest.importorskip("oracle.c_oracle")


@pytest.fixture(scope="module")
def built():
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True, capture_output=True)
    return CO


@pytest.mark.parametrize("ph", [O.Phys(), O.Phys(n=3.3, C=7e-8, q=1.0)])
def test_c_rhs_and_vjp_match_numpy(built, ph):
    H0, B = O.synthetic_icecap(70, 53, 100.0)
    H0 = H0 * 0.4
    law = O.Law(kind=O.LAW_CONST_A, A=2.21e-18)
    assert rel_l2(built.rhs(H0, B, 100.0, 100.0, ph, 2.21e-18), O.sia2d_rhs(H0, B, 100.0, 100.0, ph, law)) < 1e-13
    lam = np.random.default_rng(0).standard_normal(H0.shape)
    assert rel_l2(built.vjp_H(lam, H0, B, 100.0, 100.0, ph, 2.21e-18), O.vjp_H(lam, H0, B, 100.0, 100.0, ph, law)) < 1e-13


def test_c_rk_step_matches_numpy(built):
    ph = O.Phys()
    H0, B = O.synthetic_valley(64, 48, 50.0)
    law = O.Law(kind=O.LAW_CONST_A, A=1e-17)
    st = built.Stepper(H0, B, 50.0, 50.0, ph, 1e-17)
    f = lambda H: O.sia2d_rhs(H, B, 50.0, 50.0, ph, law)
    u = H0
    for _ in range(3):
        u, ut = O.rdpk3sp35_step(f, u, 2e-3)
        e = st.step(2e-3)
    assert rel_l2(st.u, u) < 1e-13


def test_c_dual_grid_A_field_matches_numpy(built):
    """The gridded-law path of the CPU baseline (A = NN_theta(T) hoisted into a dual-grid field, Laws.jl:339-358): RHS and
    three RDPK3Sp35 steps against the numpy oracle with the same field; a constant field equals the scalar path bit for bit."""
    ph = O.Phys()
    H0, B = O.synthetic_icecap(64, 50, 100.0)
    H0 = H0 * 0.5
    rng = np.random.default_rng(3)
    Af = 10.0 ** rng.uniform(-18.0, -16.5, (63, 49))
    law = O.Law(kind=O.LAW_CONST_A, A=Af)
    assert rel_l2(built.rhs(H0, B, 100.0, 100.0, ph, Af), O.sia2d_rhs(H0, B, 100.0, 100.0, ph, law)) < 1e-13
    assert np.array_equal(built.rhs(H0, B, 100.0, 100.0, ph, np.full((63, 49), 3e-17)), built.rhs(H0, B, 100.0, 100.0, ph, 3e-17))
    st = built.Stepper(H0, B, 100.0, 100.0, ph, Af)
    f = lambda H: O.sia2d_rhs(H, B, 100.0, 100.0, ph, law)
    u = H0
    for _ in range(3):
        u, _ = O.rdpk3sp35_step(f, u, 1e-3)
        st.step(1e-3)
    assert rel_l2(st.u, u) < 1e-13
    ms = built.MultiStepper(2, H0, B, 100.0, 100.0, ph, Af)
    ms.run(3, 1e-3)
    assert np.array_equal(ms.us[0], st.u) and np.array_equal(ms.us[1], st.u)
