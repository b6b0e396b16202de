"""GPU: the N > 1 path with REAL per-shard results.  Two ranks (torch.distributed.run, gloo collectives, both on device
0 -- a 1-GPU box cannot host two RCCL ranks) shard 4 ragged glaciers, each runs SIA2D_grad_b on its shard (per-glacier
classical law slots, trainable initial condition, Tikhonov regulariser looped over the rank's own glaciers) and the single
all-reduce of [loss, dtheta] must reproduce the one-rank result (SIA2D_grad!, gradient.jl:6-31; slot ownership
Model.jl:214-216).  Plus the library's own RCCL communicator (odinn_comm_*) as a single-rank group."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_two_rank_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_equal_one_rank(gpu, tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    one = str(tmp_path / "one")
    subprocess.run([sys.executable, WORKER, one], check=True, env=env, timeout=600)
    two = str(tmp_path / "two")
    env2 = dict(env, ODINN_DEVICE="0", ODINN_DIST_BACKEND="gloo")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                    "127.0.0.1", "--master-port", str(_free_port()), WORKER, two], check=True, env=env2, timeout=900)
    r1 = np.load(one + ".rank0.npz")
    ra, rb = np.load(two + ".rank0.npz"), np.load(two + ".rank1.npz")
    assert sorted(list(ra["mine"]) + list(rb["mine"])) == [0, 1, 2, 3] and len(ra["mine"]) and len(rb["mine"])
    # both ranks hold the same reduced values ...
    assert float(ra["loss"]) == float(rb["loss"]) and np.array_equal(ra["dth"], rb["dth"])
    # ... equal to the single-rank result: per-glacier slots have one owner (bit-identical there), the loss is a sum of
    # per-glacier terms in a different order
    assert abs(float(ra["loss"]) - float(r1["loss"])) <= 1e-13 * abs(float(r1["loss"]))
    assert np.linalg.norm(ra["dth"] - r1["dth"]) <= 1e-13 * np.linalg.norm(r1["dth"])
    assert np.isfinite(r1["dth"]).all() and np.linalg.norm(r1["dth"]) > 0


def test_rccl_communicator_single_rank_group(gpu):
    """odinn_comm_* and odinn_batch_loss_grad through RCCL itself (ncclCommInitRank / ncclAllReduce) with a group of one
    rank -- the code path the 8-GPU runs take, minus the peers."""
    from oracle import sia2d_oracle as O

    uid = gpu.Comm.unique_id()
    assert len(uid) == gpu.Comm.ID_BYTES
    comm = gpu.Comm(0, 1, 0, uid)
    x = np.arange(84, dtype=float) * 0.37 - 3.0
    assert np.array_equal(comm.allreduce_sum(x), x)
    nx, ny = 96, 80
    H0, B = O.synthetic_valley(nx, ny, 50.0)
    ts = [2010.0 + j / 12.0 for j in range(4)]
    b = gpu.GlacierBatch([(nx, ny)], [50.0], A=[4e-17])
    b.set_fields(0, H0, B)
    b.set_reference(0, ts, [H0 * (1.0 - 0.03 * j) for j in range(len(ts))], 3)
    L0, g0 = b.loss_grad(ts, reltol=1e-8)
    L1, g1 = b.batch_loss_grad(comm, ts, reltol=1e-8)
    L2, g2 = b.batch_loss_grad(None, ts, reltol=1e-8)
    assert L0 == L1 == L2 and np.array_equal(g0, g1) and np.array_equal(g0, g2)
    Lc0, gc0 = b.loss_grad_continuous(ts, reltol=1e-8, n_quadrature=12)
    Lc1, gc1 = b.batch_loss_grad(comm, ts, continuous=True, n_quadrature=12, reltol=1e-8)
    assert Lc0 == Lc1 and np.array_equal(gc0, gc1)
    b.close()
    comm.close()
