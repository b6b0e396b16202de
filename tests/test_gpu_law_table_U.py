"""GPU: the tabulated U law (law mode LM_UTAB): LawU's inputs are Hbar and |grad S| (Laws.jl:97-183) -- one bivariate function for
the whole batch, read by the stencil kernels of the solve and of both adjoints from bi-quintic patches (16 x 8 ... 128 x 64, the coarsest resolution that passes) built from the
network (and checked against it to 1e-12) unless odinn_schedule.law_table = 0.  Same contract as the Y law's table
(test_gpu_law_table.py): table == network to 1e-11 (states) / 1e-7 (gradients: central differences with 1e-4 and 1e-6 amplify any
difference in U), oracle tolerances of the network path, overflow repeats the solve, the seams keep the network."""
import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays, sched_env
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_overrides(monkeypatch):
    sched_env(monkeypatch, LAW_TABLE=None)


def _batch(gpu, arch="default", shapes=((56, 40), (70, 57))):
    from test_gpu_parity import _mlp_pair

    widths, acts = {"light": ([2, 3, 1], [1, 2]), "default": ([2, 3, 10, 3, 1], [1, 1, 1, 2]), "runtime": ([2, 5, 10, 5, 1], [3, 3, 3, 1])}[arch]
    om, gm, th = _mlp_pair(gpu, widths, acts, [(0.0, 300.0), (0.0, 0.5)], O.POST_EXPMAX, 0.0, 50.0)
    b = gpu.GlacierBatch(list(shapes), [50.0] * len(shapes))
    fields = []
    for g, (nx, ny) in enumerate(shapes):
        H0, B = O.synthetic_alpine(nx, ny, hmax=150.0 + 30.0 * g, slope=0.1)
        b.set_fields(g, H0, B)
        fields.append((H0, B))
    b.set_law(gpu.LAW_NN_U, gm, th)
    return b, om, th, fields


@pytest.mark.parametrize("arch", ["light", "default", "runtime"])
def test_table_reproduces_the_network(gpu, arch):
    b, om, th, fields = _batch(gpu, arch)
    ts = [2010.0 + j / 24.0 for j in range(4)]
    b.set_schedule(law_table=0)
    assert not b.law_table()["usable"]
    b.solve(ts, reltol=1e-8)
    Hn = [b.snapshot(g, 3) for g in range(2)]
    for g in range(2):
        b.set_reference(g, ts, [fields[g][0] * (1.0 - 0.01 * j) for j in range(4)], 3)
    Ln, gn = b.loss_grad(ts, theta=th, reltol=1e-8)
    Lc, gc = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
    b.set_schedule()
    info = b.law_table()
    assert info["usable"] and info["max_rel_dev"] < 1e-12 and info["n_intervals"] in (16 * 8, 32 * 16, 64 * 32, 128 * 64), info
    b.solve(ts, reltol=1e-8)
    for g in range(2):
        assert rel_l2(b.snapshot(g, 3), Hn[g]) < 1e-11
    Lt, gt = b.loss_grad(ts, theta=th, reltol=1e-8)
    assert abs(Lt - Ln) <= 1e-10 * abs(Ln) and rel_l2(gt, gn) < 1e-7, (Lt, Ln, rel_l2(gt, gn))
    Ltc, gtc = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
    assert abs(Ltc - Lc) <= 1e-10 * abs(Lc) and rel_l2(gtc, gc) < 1e-6, (Ltc, Lc, rel_l2(gtc, gc))
    lam = np.random.default_rng(5).standard_normal(fields[0][0].shape)
    v1, d1 = b.vjp_H(0, lam, fields[0][0]), b.dhdt(0, fields[0][0])
    b.set_schedule(law_table=0)
    assert np.array_equal(b.vjp_H(0, lam, fields[0][0]), v1) and np.array_equal(b.dhdt(0, fields[0][0]), d1)  # seams: the network
    b.close()


def test_table_path_against_the_oracle(gpu):
    b, om, th, fields = _batch(gpu, "default", ((56, 40),))
    H0, B = fields[0]
    ph = O.Phys()
    ts = [2010.0 + j / 48.0 for j in range(4)]
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    law = O.Law(kind=O.LAW_NN_U, mlp=om, theta=th)
    cfg = O.SimConfig(tstops=ts, reltol=1e-8)
    ref, _, _ = O.forward(gl, law, cfg)
    assert b.law_table()["usable"]
    b.solve(ts, reltol=1e-8)
    assert rel_l2(b.snapshot(0, 3), ref[3]) < 1e-6
    ref = [r * (1.0 + 0.02 * j) for j, r in enumerate(ref)]
    b.set_reference(0, ts, ref, 3)
    Lo, go, lam0, _ = O.loss_and_grad_continuous(gl, law, cfg, ref, ts, O.ContinuousAdjointCfg(n_quadrature=8))
    Lg, gg = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
    assert abs(Lg - Lo) <= 1e-6 * abs(Lo)
    ratio, angle, relerr = stats_err_arrays(gg, go)
    assert abs(ratio) < 2e-4 and relerr < 2e-4, (ratio, angle, relerr)
    assert rel_l2(b.lambda0(0), lam0) < 2e-4
    b.close()


def test_overflow_widens_the_table(gpu, monkeypatch):
    monkeypatch.setenv("ODINN_LAW_TABLE_HMAX", "60")  # ice up to 180 m: 60 -> 120 -> 240
    b, om, th, fields = _batch(gpu, "default")
    ts = [2010.0 + j / 24.0 for j in range(3)]
    b.set_schedule(law_table=0)
    b.solve(ts, reltol=1e-8)
    Hn = b.snapshot(1, 2)
    b.set_schedule()
    assert b.law_table()["hmax"][0] == 60.0
    b.solve(ts, reltol=1e-8)
    assert b.law_table()["hmax"][0] == 240.0 and b.law_table()["usable"]
    assert rel_l2(b.snapshot(1, 2), Hn) < 1e-11
    b.close()


@pytest.mark.parametrize("rows", ["7", "4", "2"])
def test_fused_reverse_step_of_the_U_table_matches_the_staged_one(gpu, monkeypatch, rows):
    """The ContinuousAdjoint's reverse step of target :D through the table as ONE kernel (k_adj_fused_strip<..., UT>: D = Hbar U,
    alpha and beta by the reference's central differences on the node's bi-quintic patch, face form of the H-VJP) against the five
    k_adj_stage<LM_UTAB> launches (ODINN_ADJ_UT_FUSED=0): same loss, reverse step counts within the accept-threshold flips, gradient
    and lambda(t0) to the tolerance of the adaptive reverse solve; ragged batch, three tile heights."""
    sched_env(monkeypatch, ADJ_ROWS=rows)
    out = {}
    for mode in ("0", "1"):
        sched_env(monkeypatch, ADJ_UT_FUSED=mode)
        b, om, th, fields = _batch(gpu, "default", ((56, 40), (70, 57), (131, 64)))
        ts = [2010.0 + j / 24.0 for j in range(4)]
        for g in range(3):
            b.set_reference(g, ts, [fields[g][0] * (1.0 - 0.01 * j) for j in range(4)], 3)
        assert b.law_table()["usable"]
        L, g_ = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
        out[mode] = (L, np.array(g_, dtype=float).ravel(), [b.lambda0(k) for k in range(3)], [(s.naccept, s.nreject) for s in b.last_stats_rev])
        b.close()
    a, f = out["0"], out["1"]
    assert abs(a[0] - f[0]) <= 1e-13 * abs(a[0])  # (the loss partials are summed per tile of the kernel that meets the stop: strips vs 64 x 16 tiles)
    for (na, ra), (nf, rf) in zip(a[3], f[3]):
        assert abs(na - nf) <= max(2, na // 10) and abs(ra - rf) <= max(2, ra // 2), (a[3], f[3])
    assert np.linalg.norm(a[1] - f[1]) <= 2e-6 * np.linalg.norm(a[1]), np.linalg.norm(a[1] - f[1]) / np.linalg.norm(a[1])
    for la, lf in zip(a[2], f[2]):
        assert rel_l2(lf, la) < 2e-6


@pytest.mark.parametrize("skip", [1, 0])
def test_lds_tile_reverse_step_of_the_U_table_matches_the_staged_one(gpu, monkeypatch, skip):
    """The reverse step fused over LDS TILES (k_adj_fused_lds<LM_UTAB>, odinn_schedule.adj_ut_fused = 2: one dual node per thread at a
    time with vjpH_node<LM_UTAB> -- the staged kernels' own node function -- on 54 x 22 tiles whose regions shrink by a ring per
    stage) against the five k_adj_stage<LM_UTAB> launches (adj_ut_fused = 0): the nodes and cells see the same expressions, only the
    error partials are summed over different tiles, so the loss agrees to 1e-13, the reverse step counts to the accept-threshold flips
    and gradient / lambda(t0) to the tolerance of the adaptive reverse solve; ragged batch (a glacier narrower than a tile, one wider than
    two), with and without the ice-free shortcut.  Reference: target_D_pure.jl:78-137, gradient.jl:316-324."""
    sched_env(monkeypatch, ADJ_UT_FUSED=None)
    out = {}
    for mode in (0, 2):
        b, om, th, fields = _batch(gpu, "default", ((56, 40), (70, 57), (131, 64)))
        b.set_schedule(adj_ut_fused=mode, adj_skip=skip)
        ts = [2010.0 + j / 24.0 for j in range(4)]
        for g in range(3):
            b.set_reference(g, ts, [fields[g][0] * (1.0 - 0.01 * j) for j in range(4)], 3)
        assert b.law_table()["usable"]
        L, g_ = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
        out[mode] = (L, np.array(g_, dtype=float).ravel(), [b.lambda0(k) for k in range(3)], [(s.naccept, s.nreject) for s in b.last_stats_rev])
        b.close()
    a, f = out[0], out[2]
    assert abs(a[0] - f[0]) <= 1e-13 * abs(a[0])
    for (na, ra), (nf, rf) in zip(a[3], f[3]):
        assert abs(na - nf) <= max(2, na // 10) and abs(ra - rf) <= max(2, ra // 2), (a[3], f[3])
    assert np.linalg.norm(a[1] - f[1]) <= 2e-6 * np.linalg.norm(a[1]), np.linalg.norm(a[1] - f[1]) / np.linalg.norm(a[1])
    for la, lf in zip(a[2], f[2]):
        assert rel_l2(lf, la) < 2e-6


def test_table_resolution_is_chosen_by_measurement(gpu, monkeypatch):
    """The table's resolution: the coarsest of 16 x 8 ... 128 x 64 patches that passes the check (the default) against the finest
    (ODINN_UTAB_LEVEL=3, the fixed size of round 4): both within 1e-12 of the network, same solve to 1e-11, same gradients to the
    tolerance the table has against the network."""
    out = {}
    for lev in (None, "3"):
        if lev is None:
            monkeypatch.delenv("ODINN_UTAB_LEVEL", raising=False)
        else:
            monkeypatch.setenv("ODINN_UTAB_LEVEL", lev)
        b, om, th, fields = _batch(gpu, "default")
        ts = [2010.0 + j / 24.0 for j in range(4)]
        for g in range(2):
            b.set_reference(g, ts, [fields[g][0] * (1.0 - 0.01 * j) for j in range(4)], 3)
        b.solve(ts, reltol=1e-8)
        info = b.law_table()
        assert info["usable"] and info["max_rel_dev"] < 1e-12, info
        L, gd = b.loss_grad(ts, theta=th, reltol=1e-8)
        Lc, gc = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
        out[lev] = (info["n_intervals"], [b.snapshot(g, 3) for g in range(2)], L, np.array(gd, float).ravel(), Lc, np.array(gc, float).ravel())
        b.close()
    a, f = out[None], out["3"]
    assert f[0] == 128 * 64 and a[0] <= f[0]
    for g in range(2):
        assert rel_l2(a[1][g], f[1][g]) < 1e-11
    assert abs(a[2] - f[2]) <= 1e-10 * abs(f[2]) and rel_l2(a[3], f[3]) < 1e-7
    assert abs(a[4] - f[4]) <= 1e-10 * abs(f[4]) and rel_l2(a[5], f[5]) < 1e-6


@pytest.mark.parametrize("vjp", ["discrete", "continuous"])
def test_staged_patches_are_bit_identical_to_global_loads(gpu, monkeypatch, vjp):
    """The reverse tile kernels stage the patches of their tile's nodes in LDS (utab_stage: patch rectangle of the workgroup, every
    node's patch and patch coordinates from one pass); ODINN_UT_LDS=0 keeps the same evaluation on loads from the global table.  Same
    arithmetic on the same coefficients: bit-identical losses, gradients and lambda(t0), ragged batch, both VJP stencils."""
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ODINN_UT_LDS", mode)
        b, om, th, fields = _batch(gpu, "default", ((56, 40), (70, 57), (131, 64)))
        if vjp == "continuous":
            b.set_vjp_method(gpu._lib.VJP_CONTINUOUS)
        ts = [2010.0 + j / 24.0 for j in range(4)]
        for g in range(3):
            b.set_reference(g, ts, [fields[g][0] * (1.0 - 0.01 * j) for j in range(4)], 3)
        assert b.law_table()["usable"]
        L, gd = b.loss_grad(ts, theta=th, reltol=1e-8)
        Lc, gc = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
        out[mode] = (L, np.array(gd, float).ravel(), Lc, np.array(gc, float).ravel(), [b.lambda0(k) for k in range(3)])
        b.close()
    a, f = out["1"], out["0"]
    assert a[0] == f[0] and np.array_equal(a[1], f[1])
    assert a[2] == f[2] and np.array_equal(a[3], f[3])
    for la, lf in zip(a[4], f[4]):
        assert np.array_equal(la, lf)
