"""GPU: the tabulated Y law (odinn_schedule.law_table, law mode LM_YTAB).  LawY's inputs are the glacier's scalar temperature
and Hbar (Laws.jl:240-273), so per glacier and theta the law is a function of one variable; unless law_table = 0 the stencil
kernels of the forward solve and of both adjoints read it from a table of quintics built from the network itself.  The table
path must reproduce the network path -- forward states to 1e-11, gradients to 1e-8 (the reference's finite-difference
partial of the law, target_D_hybrid.jl:58-71, amplifies any difference in Y by 1e4) -- stay within the oracle tolerances of the
network path, follow theta, and survive a solve that leaves its range."""
import numpy as np
import pytest

from conftest import rel_l2, stats_err_arrays, sched_env
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_overrides(monkeypatch):  # (the suite is also run under ODINN_LAW_TABLE=1 / ODINN_INTERP_ASYNC=0: these tests set the fields)
    sched_env(monkeypatch, LAW_TABLE=None)
    sched_env(monkeypatch, INTERP_ASYNC=None)

ARCHS = {"light": ([2, 3, 1], [1, 2]), "default": ([2, 3, 10, 3, 1], [1, 1, 1, 2]), "x16": ([2, 16, 16, 1], [1, 1, 2]),
         "runtime": ([2, 5, 10, 5, 1], [3, 3, 3, 1])}


def _batch(gpu, arch, shapes=((56, 40),), Ts=(-5.0,), seed=3):
    from test_gpu_parity import _mlp_pair

    ph = O.Phys()
    widths, acts = ARCHS[arch]
    om, gm, th = _mlp_pair(gpu, widths, acts, [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
    b = gpu.GlacierBatch(list(shapes), [50.0] * len(shapes), T=list(Ts))
    fields = []
    for g, (nx, ny) in enumerate(shapes):
        H0, B = O.synthetic_alpine(nx, ny, hmax=150.0 + 30.0 * g, slope=0.1)
        b.set_fields(g, H0, B)
        fields.append((H0, B))
    b.set_law(gpu.LAW_NN_Y, gm, th)
    return b, om, th, fields, ph


@pytest.mark.parametrize("arch", list(ARCHS))
def test_table_reproduces_the_network_in_the_solve_and_in_both_gradients(gpu, arch):
    shapes, Ts = ((56, 40), (70, 57)), (-5.0, -11.0)
    b, om, th, fields, ph = _batch(gpu, arch, shapes, Ts)
    ts = [2010.0 + j / 24.0 for j in range(4)]
    b.set_schedule(law_table=0)
    b.solve(ts, reltol=1e-8)
    Hn = [b.snapshot(g, 3) for g in range(2)]
    for g in range(2):
        b.set_reference(g, ts, [fields[g][0] * (1.0 - 0.01 * j) for j in range(4)], 3)
    Ln, gn = b.loss_grad(ts, theta=th, reltol=1e-8)
    Lc, gc = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
    assert not b.law_table()["usable"]
    b.set_schedule()  # automatic: the table
    info = b.law_table()
    assert info["usable"] and info["max_rel_dev"] < 1e-12 and (info["hmax"] >= 200.0).all(), info
    st = b.solve(ts, reltol=1e-8)
    for g in range(2):
        assert rel_l2(b.snapshot(g, 3), Hn[g]) < 1e-11
    Lt, gt = b.loss_grad(ts, theta=th, reltol=1e-8)
    assert abs(Lt - Ln) <= 1e-10 * abs(Ln) and rel_l2(gt, gn) < 1e-8, (Lt, Ln, rel_l2(gt, gn))
    Ltc, gtc = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
    assert abs(Ltc - Lc) <= 1e-10 * abs(Lc) and rel_l2(gtc, gc) < 1e-7, (Ltc, Lc, rel_l2(gtc, gc))
    # the table follows theta (rebuilt by the gradient call) ...
    th2 = th * 1.03
    Lt2, gt2 = b.loss_grad(ts, theta=th2, reltol=1e-8)
    b.set_schedule(law_table=0)
    Ln2, gn2 = b.loss_grad(ts, theta=th2, reltol=1e-8)
    # (5e-8: the forward solves of the two runs are different kernels -- strip with the table, LDS tiles with the network -- and the
    #  1e-4 forward difference inside the Y law's H-VJP turns their 1e-12 into 1e-8 over the reverse-Euler loop; seen 1.5e-8)
    assert Ln2 != Ln and abs(Lt2 - Ln2) <= 1e-10 * abs(Ln2) and rel_l2(gt2, gn2) < 5e-8
    # ... and the seams never use it (arbitrary fields from the caller): bit-identical with the schedule on or off
    lam = np.random.default_rng(5).standard_normal(shapes[0])
    v0, d0 = b.vjp_H(0, lam, fields[0][0]), b.dhdt(0, fields[0][0])
    b.set_schedule()
    assert b.law_table()["usable"]
    assert np.array_equal(b.vjp_H(0, lam, fields[0][0]), v0) and np.array_equal(b.dhdt(0, fields[0][0]), d0)
    b.close()


def test_table_path_against_the_oracle(gpu):
    """The tolerances of the network path (test_continuous_adjoint_other_law_modes) hold for the table path."""
    b, om, th, fields, ph = _batch(gpu, "default")
    H0, B = fields[0]
    ts = [2010.0 + j / 48.0 for j in range(4)]
    gl = O.Glacier(H0, B, 50.0, 50.0, ph)
    law = O.Law(kind=O.LAW_NN_Y, mlp=om, theta=th, T=-5.0)
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
    Lo2, go2 = O.loss_and_grad(gl, law, cfg, ref, ts)[:2]
    Ld, gd = b.loss_grad(ts, theta=th, reltol=1e-8)
    assert abs(Ld - Lo2) <= 1e-6 * abs(Lo2)
    ratio, angle, relerr = stats_err_arrays(gd, go2)
    assert abs(ratio) < 1e-5 and relerr < 1e-5, (ratio, angle, relerr)
    b.close()


def test_a_solve_that_leaves_the_table_is_repeated_with_a_wider_one(gpu, monkeypatch):
    monkeypatch.setenv("ODINN_LAW_TABLE_HMAX", "40")  # the ice is up to 150 m thick: two widenings (40 -> 80 -> 160)
    b, om, th, fields, ph = _batch(gpu, "default")
    ts = [2010.0 + j / 24.0 for j in range(3)]
    b.set_schedule(law_table=0)
    b.solve(ts, reltol=1e-8)
    Hn = b.snapshot(0, 2)
    b.set_reference(0, ts, [fields[0][0]] * 3, 3)
    Ln, gn = b.loss_grad(ts, theta=th, reltol=1e-8)
    b.set_schedule()
    assert b.law_table()["hmax"][0] == 40.0
    b.solve(ts, reltol=1e-8)
    assert b.law_table()["hmax"][0] == 160.0 and b.law_table()["usable"]
    assert rel_l2(b.snapshot(0, 2), Hn) < 1e-11
    Lt, gt = b.loss_grad(ts, theta=th, reltol=1e-8)
    assert abs(Lt - Ln) <= 1e-10 * abs(Ln) and rel_l2(gt, gn) < 1e-8
    # a range that two doublings cannot reach: the batch falls back to the network until its fields change
    monkeypatch.setenv("ODINN_LAW_TABLE_HMAX", "5")
    b.set_fields(0, *fields[0])
    b.solve(ts, reltol=1e-8)
    assert not b.law_table()["usable"]
    assert np.array_equal(b.snapshot(0, 2), Hn)
    b.close()


def test_a_table_that_misses_the_tolerance_is_not_used(gpu):
    """A network with huge first-layer weights oscillates inside one table interval: the build's own check keeps the network."""
    b, om, th, fields, ph = _batch(gpu, "light")
    th2 = th.copy()  # 2 -> 3 -> 1: every hidden unit softplus(4000 (Hbar_norm + 0.3)), a kink at Hbar = 100 m one interval wide
    th2[0:3], th2[3:6], th2[6:9] = 0.0, 4000.0, 1200.0
    b.set_theta(th2)
    info = b.law_table()
    assert not info["usable"] and info["max_rel_dev"] > 1e-12, info
    ts = [2010.0, 2010.02]
    b.solve(ts, reltol=1e-8)
    H1 = b.snapshot(0, 1)
    b.set_schedule(law_table=0)
    b.solve(ts, reltol=1e-8)
    assert np.array_equal(b.snapshot(0, 1), H1)
    b.close()


@pytest.mark.parametrize("select", ["1", "0"])
@pytest.mark.parametrize("table", [0, 1])
def test_overlapped_interpolation_is_bit_identical(gpu, monkeypatch, table, select):
    """Both adjoints of the Y law: the `:Linear` contraction of a stop on lane streams, overlapped with the following reverse
    steps (odinn_schedule.interp_async: the default 3 lanes, 1, 4), gives the same bits whatever the number of lanes -- every
    contribution has its own slot, the slots are added in the order of the stops.  With the radix-sorted contraction
    (ODINN_INTERP_SELECT=0) that is also the sequence on the batch's own stream (interp_async = 0), bit for bit; the sort-free
    contraction of the lanes (the default: fixed-point interval sums, and with the table the fused step's own emission of the
    node pairs) differs from that sequence in the rounding of its sums only."""
    monkeypatch.setenv("ODINN_INTERP_SELECT", select)
    shapes, Ts = ((56, 40), (70, 57), (131, 64)), (-5.0, -11.0, -2.0)
    b, om, th, fields, ph = _batch(gpu, "default", shapes, Ts)
    ts = [2010.0 + j / 24.0 for j in range(4)]
    for g in range(3):
        b.set_reference(g, ts, [fields[g][0] * (1.0 - 0.01 * j) for j in range(4)], 3)
    assert b.get_schedule()["interp_async"] != 0
    res = {}
    for mode in (-1, 0, 1, 4, -1):
        b.set_schedule(law_table=-1 if table else 0, interp_async=mode)
        res.setdefault(mode, []).append((b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=24),
                                         b.loss_grad(ts, theta=th, reltol=1e-8)))
    (Lc, gc), (Ld, gd) = res[0][0]
    assert np.isfinite(gc).all() and np.linalg.norm(gc) > 0 and np.isfinite(gd).all() and np.linalg.norm(gd) > 0
    (_, gcl), (_, gdl) = res[1][0]
    for mode, runs in res.items():
        for (Lc2, gc2), (Ld2, gd2) in runs:
            assert Lc2 == Lc and Ld2 == Ld, mode
            if select == "0":
                assert np.array_equal(gc2, gc) and np.array_equal(gd2, gd), mode
            elif mode != 0:
                assert np.array_equal(gc2, gcl) and np.array_equal(gd2, gdl), mode
                assert rel_l2(gc2, gc) < 1e-11 and rel_l2(gd2, gd) < 1e-11, mode
    b.close()


def test_fused_reverse_step_with_the_table_matches_the_staged_one(gpu):
    """n_H = n_gradS = 3 without sliding: the Y law is the integer-power law with Y(Hbar) in A's place, and the ContinuousAdjoint's
    reverse step runs as ONE k_adj_fused_strip<..., YT> launch instead of five k_adj_stage<., LM_YTAB> (adj_fused = 0): same step
    counts, gradients equal to the rounding of two compilations of the same expressions; 4- and 7-row tiles, with and without the
    interleaved snapshot pairs."""
    shapes, Ts = ((70, 57), (131, 64)), (-5.0, -9.0)
    b, om, th, fields, ph = _batch(gpu, "default", shapes, Ts)
    ts = [2010.0 + j / 24.0 for j in range(4)]
    for g in range(2):
        b.set_reference(g, ts, [fields[g][0] * (1.0 - 0.01 * j) for j in range(4)], 3)
    b.set_schedule(adj_fused=0)
    Ls, gs = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=16)
    rev_s = [(s.naccept, s.nreject) for s in b.last_stats_rev]
    for sched in (dict(), dict(adj_rows=4), dict(adj_rows=7), dict(adj_rows=7, adj_segs=0), dict(adj_skip=0)):
        b.set_schedule(**sched)
        Lf, gf = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=16)
        assert [(s.naccept, s.nreject) for s in b.last_stats_rev] == rev_s, sched
        assert abs(Lf - Ls) <= 1e-12 * abs(Ls) and rel_l2(gf, gs) < 1e-9, (sched, rel_l2(gf, gs))
    b.set_schedule(law_table=0)  # the network: five stage launches, as before
    Ln, gn = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=16)
    assert rel_l2(gf, gn) < 1e-7
    b.close()


def test_contraction_over_the_active_nodes_matches_the_dense_one(gpu):
    """The overlapped `:Linear` contractions of a gradient sort / gather / sum only the dual nodes that carry ice in at least one
    snapshot of its forward solve (launch_interp_active); the one-stream sequence (interp_async = 0) still runs over all dual
    nodes.  Same keys, same stable sort, the dropped nodes have Hbar = 0 and weight 0 at every stop: gradients agree to the
    rounding of the interval sums' block partition, in both adjoints, on a ragged batch with a glacier that is mostly ice-free."""
    shapes, Ts = ((56, 40), (131, 64), (70, 57)), (-5.0, -11.0, -2.0)
    b, om, th, fields, ph = _batch(gpu, "default", shapes, Ts)
    H1, B1 = fields[1]
    H1 = H1.copy(); H1[:, : H1.shape[1] // 2] = 0.0  # half of glacier 1 without ice
    b.set_fields(1, np.asfortranarray(H1), B1)
    fields[1] = (H1, B1)
    ts = [2010.0 + j / 24.0 for j in range(4)]
    for g in range(3):
        b.set_reference(g, ts, [fields[g][0] * (1.0 - 0.01 * j) for j in range(4)], 3)
    out = {}
    for mode in (0, -1):
        b.set_schedule(interp_async=mode)
        out[mode] = (b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=24), b.loss_grad(ts, theta=th, reltol=1e-8))
    for k in range(2):
        (La, ga), (Ld, gd) = out[-1][k], out[0][k]
        assert abs(La - Ld) <= 1e-13 * abs(Ld) and rel_l2(ga, gd) < 1e-12, (k, rel_l2(ga, gd))
        assert np.isfinite(ga).all() and np.linalg.norm(ga) > 0
    b.close()
