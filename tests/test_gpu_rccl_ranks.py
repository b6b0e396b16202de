"""GPU, >= 2 devices: REAL RCCL ranks, one GPU each, through the library's communicator -- the code path of the 1/2/4/8-GPU
scaling runs (SIA2D_grad!, src/inverse/SIA2D/gradient.jl:6-31; aggregate of Model.jl:208-224).  Skipped on a one-GPU box
(two RCCL ranks cannot share a device); the one-GPU coverage of the same entry points is tests/test_gpu_two_rank.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_rccl_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(n, out, case, env):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), WORKER, out, case]
    subprocess.run(cmd, check=True, env=env, timeout=900)


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "ODINN_DEVICE", "ODINN_DIST_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@pytest.mark.parametrize("nranks", [2, 4])
def test_rccl_ranks_reproduce_the_single_rank_gradient(gpu, tmp_path, nranks):
    if gpu.device_count() < nranks:
        pytest.skip(f"{nranks} RCCL ranks need {nranks} devices ({gpu.device_count()} visible)")
    env = _env()
    one = str(tmp_path / "one")
    subprocess.run([sys.executable, WORKER, one, "ok"], check=True, env=env, timeout=600)
    many = str(tmp_path / "many")
    _launch(nranks, many, "ok", env)
    r1 = np.load(one + ".rank0.npz")
    rs = [np.load(f"{many}.rank{r}.npz") for r in range(nranks)]
    assert sorted(i for r in rs for i in r["mine"]) == [0, 1, 2, 3]
    for r in rs:  # every rank returns the GLOBAL loss and gradient, bit-identical across ranks ...
        for k in ("Ld", "gd", "Lc", "gc", "x"):
            assert np.array_equal(r[k], rs[0][k]), k
        assert np.array_equal(r["x"], nranks * np.arange(5.0) + sum(range(nranks)))
        # ... equal to the one-rank result up to the order of the sums over glaciers
        assert abs(float(r["Ld"]) - float(r1["Ld"])) <= 1e-12 * abs(float(r1["Ld"]))
        assert np.linalg.norm(r["gd"] - r1["gd"]) <= 1e-12 * np.linalg.norm(r1["gd"])
        assert abs(float(r["Lc"]) - float(r1["Lc"])) <= 1e-12 * abs(float(r1["Lc"]))
        assert np.linalg.norm(r["gc"] - r1["gc"]) <= 1e-12 * np.linalg.norm(r1["gc"])


def test_rank_local_failure_reaches_every_rank(gpu, tmp_path):
    """maxiters exhausted on rank 1 only: odinn_batch_loss_grad's status slot (the first entry of the all-reduced vector) makes
    both ranks fail together, nobody hangs, and the next call succeeds."""
    if gpu.device_count() < 2:
        pytest.skip(f"2 RCCL ranks need 2 devices ({gpu.device_count()} visible)")
    out = str(tmp_path / "fail")
    _launch(2, out, "fail", _env())
    r0, r1 = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    assert "rank" in str(r0["err"]) and str(r0["err"]) != "" and str(r1["err"]) != ""
    assert float(r0["L2"]) == float(r1["L2"]) and np.array_equal(r0["g2"], r1["g2"]) and np.isfinite(r0["g2"]).all()
