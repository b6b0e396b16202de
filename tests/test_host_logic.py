"""-m "not gpu": host-side logic of the reference-facing layer (no device needed)."""
import numpy as np
import pytest

from oracle import sia2d_oracle as O


def test_shard_glaciers_balanced_and_deterministic(odinn):
    cells = [192 * 160, 96 * 80, 128 * 112, 160 * 128, 1024 * 1024, 512 * 512, 64 * 48, 300 * 200]
    for world in (1, 2, 4, 8):
        sh = odinn.shard_glaciers(cells, world)
        assert sorted(i for s in sh for i in s) == list(range(len(cells)))
        assert sh == odinn.shard_glaciers(cells, world)
    sh = odinn.shard_glaciers([100] * 64, 8)
    assert all(len(s) == 8 for s in sh)
    loads = [sum(cells[i] for i in s) for s in odinn.shard_glaciers(cells, 2)]
    assert max(loads) <= sum(cells) - min(loads) and max(loads) - 1024 * 1024 < sum(cells) / 2


def test_define_callback_steps_and_tstops(odinn):
    ts = odinn.define_callback_steps((2010.0, 2012.0), 1.0 / 12.0)
    assert len(ts) == 25 and ts[0] == 2010.0 and ts[-1] == 2012.0  # k = 25 for 2 yr monthly
    p = odinn.Parameters(simulation=odinn.SimulationParameters(tspan=(2010.0, 2011.0)))
    H = np.ones((8, 8))
    g = odinn.Glacier2D("RGI60-00.00001", H, H, 50.0, 50.0, thicknessData=odinn.ThicknessData([2010.0, 2010.5, 2011.0], [H, H, H]))
    sim = odinn.Prediction(odinn.Model(odinn.SIA2Dmodel(p)), [g], p)
    t = sim.tstops()
    assert t == sorted(set(t)) and 2010.5 in t and len(t) == 13
    assert sim.mb_times() == []


def test_default_architectures_and_theta_layout(odinn):
    """83 parameters for the default 1-input net, 86 for 2 inputs, 10/13 in test_mode
    (SURVEY 8(a) a3); theta = [vec(W) column-major, b] per layer."""
    p = odinn.Parameters()
    assert odinn.NeuralNetwork(p).n_params == 83
    p2 = odinn.Parameters(UDE=odinn.UDEparameters(target="D_hybrid"))
    assert odinn.NeuralNetwork(p2).n_params == 86
    pt = odinn.Parameters(simulation=odinn.SimulationParameters(test_mode=True))
    assert odinn.NeuralNetwork(pt).n_params == 10
    nn = odinn.NeuralNetwork(p, seed=1)
    mlp = O.MLP(nn.widths, nn.acts)
    (W1, b1), *_ = mlp.unpack(nn.theta)
    assert W1.shape == (3, 1) and np.array_equal(W1[:, 0], nn.theta[:3]) and np.all(b1 == 0)
    law = odinn.LawA(nn, p)
    assert law.mlp.post_kind == odinn.POST_AFFINE and law.mlp.post_lo == 8e-21 and law.mlp.post_hi == 8e-17
    d = law.mlp.c_struct()
    assert d.n_layers == 4 and list(d.widths)[:5] == [1, 3, 10, 3, 1]


def test_lawY_lawU_descriptors(odinn):
    p = odinn.Parameters(UDE=odinn.UDEparameters(target="D_hybrid"))
    nn = odinn.NeuralNetwork(p)
    y = odinn.LawY(nn, p)
    assert y.kind == odinn.LAW_NN_Y and y.mlp.prescale == ((-25.0, 0.0), (0.0, 500.0)) and y.mlp.post_hi == 8e-17
    u = odinn.LawU(nn, p, max_NN=50.0, prescale_bounds=[(0.0, 300.0), (0.0, 0.5)])
    assert u.kind == odinn.LAW_NN_U and u.mlp.post_kind == odinn.POST_EXPMAX
    with pytest.raises(ValueError):
        odinn.SIA2Dmodel(p, A=odinn.ConstantA(), Y=y)


def test_functional_inversion_alias(odinn):
    assert odinn.FunctionalInversion is odinn.Inversion


def test_adam_update_rule_matches_optimisers(odinn):
    """One Adam step as Optimisers.Adam: theta -= eta * mhat / (sqrt(vhat) + eps)."""
    a = odinn.Adam(0.1)
    g = np.array([1.0, -2.0])
    m = (1 - a.beta[0]) * g
    v = (1 - a.beta[1]) * g * g
    step = a.eta * (m / (1 - a.beta[0])) / (np.sqrt(v / (1 - a.beta[1])) + a.eps)
    assert np.allclose(step, 0.1 * np.sign(g), rtol=1e-6)


def test_initial_condition_and_multiloss_host_logic(odinn):
    """θ layout (law parameters, then one H0 matrix per glacier), filters equal to the oracle's
    restatement, MultiLoss bookkeeping (MultiLoss.jl:22-35)."""
    H0, B = O.synthetic_valley(20, 16, 50.0)
    H1, B1 = O.synthetic_valley(24, 12, 50.0)
    gl = [odinn.Glacier2D("a", H0, B, 50.0, 50.0), odinn.Glacier2D("b", H1, B1, 50.0, 50.0)]
    p = odinn.Parameters()
    ic = odinn.InitialCondition(p, gl)
    assert ic.sizes == [320, 288] and ic.n_params == 608
    nn = odinn.NeuralNetwork(p)
    m = odinn.Model(odinn.SIA2Dmodel(p, A=odinn.LawA(nn, p)), regressors={"A": nn, "IC": ic})
    assert m.n_main == nn.n_params and m.theta.size == nn.n_params + 608
    assert np.array_equal(m.theta[m.n_main:m.n_main + 320], H0.ravel(order="F"))
    x = np.random.default_rng(0).uniform(-3, 3, 320)
    for f in ("identity", "softplus", "Zang1980"):
        assert np.array_equal(odinn.evaluate_H0(x, gl[0], f), O.evaluate_H0(x.reshape((20, 16), order="F"), gl[0].mask, f))
        assert np.array_equal(odinn.evaluate_dH0(x, gl[0], f), O.evaluate_dH0(x.reshape((20, 16), order="F"), gl[0].mask, f))
    with pytest.raises(ValueError):
        odinn.MultiLoss(losses=(odinn.LossH(),), lambdas=(1.0, 2.0))
    from odinn_jl_amd.api import _split_loss

    d, w, regs = _split_loss(odinn.MultiLoss(losses=(odinn.LossH(), odinn.InitialThicknessRegularization(t0=2010.0)),
                                             lambdas=(2.0, 1e-3)))
    assert isinstance(d, odinn.LossH) and w == 2.0 and len(regs) == 1 and regs[0][1] == 1e-3
    with pytest.raises(ValueError):
        odinn.TikhonovRegularization(operator="gradient")
    with pytest.raises(ValueError):
        odinn.ContinuousAdjoint(interpolation="Cubic")


def test_training_result_file_and_scalar_log(odinn, tmp_path):
    """callback_diagnosis + save_inversion_file! record (callback_utils.jl:60-110,
    trainingresult_utils.jl:4-33): θ, θ_hist, ∇θ_hist, losses, params; scalar tags of the logger."""
    import json
    import types

    p = odinn.Parameters()
    sim = types.SimpleNamespace(stats=odinn.TrainingStats(), parameters=p)
    log = odinn.ScalarLogger(str(tmp_path / "run" / "scalars.jsonl"))
    th = np.arange(5.0)
    for it in range(3):
        odinn.callback_diagnosis(th + it, 10.0 / (it + 1), np.full(5, 0.5 * it), sim, save=(it == 2), tbLogger=log,
                                 path=str(tmp_path))
    log.close()
    res = odinn.load_inversion_file(str(tmp_path / "_inversion_result.npz"))
    assert np.array_equal(res.θ, th + 2) and len(res.θ_hist) == 3 and len(res.grad_hist) == 3
    assert res.losses == [10.0, 5.0, 10.0 / 3] and np.array_equal(res.grad_hist[2], np.full(5, 1.0))
    assert res.params["solver"]["reltol"] == p.solver.reltol and res.params["UDE"]["grad"]["__type__"] == "ContinuousAdjoint"
    rows = [json.loads(l) for l in open(tmp_path / "run" / "scalars.jsonl")]
    assert [r["tag"] for r in rows[:2]] == ["train/loss", "train/norm_grad"] and rows[0]["step"] == 1
    assert sum(r["tag"] == "train/time_per_iter" for r in rows) == 2  # not on the first call (callback_utils.jl:93)


def test_tensorboard_event_file(odinn, tmp_path):
    """TBLogger writes what TensorBoardLogger.jl writes for callback_diagnosis (callback_utils.jl:84-98): a tfevents file
    whose records carry masked CRC-32C checksums (known answer: crc32c("123456789") = 0xE3069283), first record
    `brain.Event:2`, then one scalar per log_value with the reference's tags and steps."""
    import glob
    import struct
    import types

    _crc32c, _masked_crc = odinn.api._crc32c, odinn.api._masked_crc

    assert _crc32c(b"123456789") == 0xE3069283 and _crc32c(b"") == 0
    assert _masked_crc(b"123456789") == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xA282EAD8) & 0xFFFFFFFF)
    sim = types.SimpleNamespace(stats=odinn.TrainingStats(), parameters=odinn.Parameters())
    log = odinn.TBLogger(str(tmp_path / "tb"))
    for it in range(3):
        odinn.callback_diagnosis(np.arange(4.0), 8.0 / (it + 1), np.full(4, 2.0), sim, tbLogger=log)
    log.log_value("train/loss", 1e-3, 300)  # a two-byte varint step
    log.close()
    files = glob.glob(str(tmp_path / "tb" / "events.out.tfevents.*"))
    assert len(files) == 1
    raw = open(files[0], "rb").read()
    (n0,) = struct.unpack("<Q", raw[:8])
    assert b"brain.Event:2" in raw[12:12 + n0]
    rows = odinn.read_event_file(files[0])
    assert [r[0] for r in rows[:2]] == ["train/loss", "train/norm_grad"] and rows[0][1] == 1
    assert sum(r[0] == "train/time_per_iter" for r in rows) == 2
    losses = [r for r in rows if r[0] == "train/loss"]
    assert [r[1] for r in losses] == [1, 2, 3, 300]
    assert np.allclose([r[2] for r in losses], [8.0, 4.0, 8.0 / 3, 1e-3], rtol=1e-7)
    assert np.isclose([r[2] for r in rows if r[0] == "train/norm_grad"][0], 4.0)
    bad = bytearray(raw)
    bad[-6] ^= 1  # a flipped payload bit is caught by the data CRC
    open(files[0], "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="CRC"):
        odinn.read_event_file(files[0])


def test_load_gridded_glacier_reads_oggm_style_files(odinn, tmp_path):
    """Real-glacier ingestion seam: an OGGM-style gridded file (NetCDF-3 through scipy, or .npz):
    (y, x) variables with y descending are returned as [x, y] with both axes increasing, thickness masked by
    glacier_mask and NaN -> 0, bed = surface - thickness."""
    from scipy.io import netcdf_file

    nx, ny, d = 7, 5, 50.0
    x = 1000.0 + d * np.arange(nx)
    y = 9000.0 - d * np.arange(ny)  # OGGM: y descending
    X, Y = np.meshgrid(x, y)  # (ny, nx)
    topo = 2000.0 + 0.1 * (X - x[0]) + 0.05 * (Y - y[-1])
    thick = np.where((X - x[3]) ** 2 + (Y - y[2]) ** 2 < (2.2 * d) ** 2, 80.0, np.nan)
    mask = np.isfinite(thick).astype("i4")
    mask[2, 3] = 0  # a cell the outline excludes although a thickness is given
    path = tmp_path / "gridded_data.nc"
    with netcdf_file(str(path), "w") as nc:
        nc.createDimension("x", nx)
        nc.createDimension("y", ny)
        for name, arr, dims, typ in (("x", x, ("x",), "d"), ("y", y, ("y",), "d"), ("topo_smoothed", topo, ("y", "x"), "d"),
                                     ("consensus_ice_thickness", thick, ("y", "x"), "d"), ("glacier_mask", mask, ("y", "x"), "i")):
            v = nc.createVariable(name, typ, dims)
            v[:] = arr
    gl = odinn.load_gridded_glacier(path, rgi_id="RGI60-11.00000", A=3e-17)
    assert gl.rgi_id == "RGI60-11.00000" and (gl.nx, gl.ny) == (nx, ny) and gl.dx == d and gl.dy == d and gl.A == 3e-17
    assert gl.H0.flags.f_contiguous and gl.B.flags.f_contiguous
    # [i, j] = [x, y] with y ascending: file row ny-1-j
    for i in range(nx):
        for j in range(ny):
            h = thick[ny - 1 - j, i]
            h = 0.0 if (not np.isfinite(h) or mask[ny - 1 - j, i] == 0) else h
            assert gl.H0[i, j] == h
            assert gl.B[i, j] == topo[ny - 1 - j, i] - h
    assert gl.H0[3, 2] == 0.0 and (gl.H0 > 0).sum() == mask.sum()
    assert gl.mask.shape == (nx, ny) and gl.mask[0, 0]
    np.savez(tmp_path / "g.npz", x=x, y=y, topo=topo, thickness=np.nan_to_num(thick), glacier_mask=mask)
    g2 = odinn.load_gridded_glacier(tmp_path / "g.npz")
    assert np.array_equal(g2.H0, gl.H0) and np.array_equal(g2.B, gl.B) and g2.rgi_id == "g"


def test_importing_the_package_leaves_the_hip_environment_alone(odinn):
    """HIP_FORCE_DEV_KERNARG is a process-wide runtime setting: neither the Python layer nor the library's loader sets it
    unless asked to (ODINN_REQUEST_DEV_KERNARG=1, and then never over a value the user has set).  Fresh interpreters."""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import _odinn_import; p = _odinn_import.load(); p._lib.lib(); import os, ctypes; "
            "g = ctypes.CDLL(None).getenv; g.restype = ctypes.c_char_p; v = g(b'HIP_FORCE_DEV_KERNARG'); "  # (the C environment: the loader's setenv does not show in os.environ)
            "print(os.environ.get('HIP_FORCE_DEV_KERNARG') if v is None else v.decode())" % root)
    for extra, want in (({}, "None"), ({"HIP_FORCE_DEV_KERNARG": "0"}, "0"), ({"ODINN_REQUEST_DEV_KERNARG": "1"}, "1"),
                        ({"ODINN_REQUEST_DEV_KERNARG": "1", "HIP_FORCE_DEV_KERNARG": "0"}, "0")):
        env = {k: v for k, v in os.environ.items() if k not in ("HIP_FORCE_DEV_KERNARG", "ODINN_REQUEST_DEV_KERNARG")}
        env.update(extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-500:]
        assert out.stdout.strip().splitlines()[-1] == want, (extra, out.stdout)


def test_library_does_not_link_rccl():
    """RCCL is resolved lazily by the odinn_comm_* entry points (dlopen): single-GPU users can load the library on a machine
    without it."""
    import os, subprocess

    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "odinn.jl_amd", "csrc", "libodinn_hip.so")
    out = subprocess.run(["readelf", "-d", so], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("readelf not available")
    needed = [l for l in out.stdout.splitlines() if "NEEDED" in l]
    assert needed and not any("rccl" in l or "nccl" in l for l in needed), needed


def test_log1p_table_recipe_is_within_two_ulp():
    """The device's table form of log1p on [0, 1] (sia2d_device.hpp: 65 entries {log1p(i/64), 1/(1 + i/64)}, degree-8
    remainder), restated with plain double arithmetic (no fma: one rounding more per step than the device), against
    60-digit arithmetic: <= 2 ulp, like the atanh-series form it replaces in most kernels."""
    import math, random
    from decimal import Decimal, getcontext

    getcontext().prec = 60
    tab = [(math.log1p(i / 64.0), 1.0 / (1.0 + i / 64.0)) for i in range(65)]

    def log1p_table(t):
        k = float(round(t * 64.0))
        e = tab[int(k)]
        r = (t - k * 0.015625) * e[1]
        p = -0.125
        for c in (1.0 / 7.0, -1.0 / 6.0, 0.2, -0.25, 1.0 / 3.0, -0.5, 1.0):
            p = p * r + c
        return r * p + e[0]

    random.seed(7)
    ts = [1.0, 0.5, 1 / 128, 1 / 64, 3 / 128, 127 / 128, 1e-17, 1e-8] + [random.random() for _ in range(12000)] + \
         [random.random() * 10 ** random.uniform(-12, 0) for _ in range(8000)]
    worst = 0.0
    for t in ts:
        ref = (Decimal(1) + Decimal(t)).ln() if t > 1e-25 else Decimal(t)
        worst = max(worst, float(abs(Decimal(log1p_table(t)) - ref) / Decimal(math.ulp(float(ref)))))
    assert log1p_table(0.0) == 0.0 and worst <= 2.0, worst


def test_trainable_components_and_inversion_binder_names(odinn):
    """Model.trainable_components (Model.jl:132-181: regressor slots by law, target, θ; splitθ :189-200) and InversionBinder
    (sciml_utils.jl:21-24) under the reference's names."""
    p = odinn.Parameters()
    nn = odinn.NeuralNetwork(p)
    H0 = np.zeros((12, 10)); H0[3:9, 3:7] = 40.0
    gl = [odinn.Glacier2D(f"g{i}", H0 * (1 + i), np.zeros_like(H0) + 1000.0, 50.0, 50.0) for i in range(2)]
    ic = odinn.InitialCondition(p, gl)
    m = odinn.Model(odinn.SIA2Dmodel(p, A=odinn.LawA(nn, p)), regressors={"A": nn, "IC": ic})
    tc = m.trainable_components
    assert tc.A is nn and tc.IC is ic and tc.Y is None and tc.U is None and isinstance(tc.target, odinn.SIA2D_A_target)
    assert tc.theta is m.theta and tc.theta.size == nn.theta.size + 2 * H0.size
    parts = tc.split_theta(m.theta, 1)
    assert np.array_equal(parts["A"], nn.theta) and np.array_equal(parts["IC"], ic.theta[H0.size:])
    inv = odinn.Inversion(m, gl, p)
    th = m.theta + 1.0
    assert odinn.InversionBinder(inv, th).apply() is inv and np.array_equal(m.theta, th)
    with pytest.raises(ValueError):
        tc.theta = np.zeros(3)
    reg = odinn.GlacierWideInv(p, gl, "A")
    m2 = odinn.Model(odinn.SIA2Dmodel(p, A=odinn.LawA(p, scalar=True)), regressors={"A": reg})
    assert m2.trainable_components.split_theta(m2.theta, 1)["A"].size == 1


def test_fuzz_env_scrub_keeps_every_variable_the_fuzz_file_reads():
    """tests/test_gpu_fuzz.py deletes ODINN_* from the environment of every draw (a developer's ODINN_SCHEDULE must not steer the
    checked schedules) -- except the variables the file itself reads.  ODINN_FUZZ_BIG was missing from that list for a round: the
    large-grid mode was dead without anybody noticing."""
    import os, re
    src = open(os.path.join(os.path.dirname(__file__), "test_gpu_fuzz.py")).read()
    read = set(re.findall(r'os\.environ\.get\("(ODINN_[A-Z_]+)"\)', src))
    keeps = re.findall(r'key not in \(([^)]*)\)', src)
    assert len(keeps) >= 5 and "ODINN_FUZZ_BIG" in read
    for k in keeps:
        kept = set(re.findall(r'"(ODINN_[A-Z_]+)"', k))
        assert read <= kept, (sorted(read - kept), k)
