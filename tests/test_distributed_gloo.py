"""-m "not gpu": the N > 1 path on CPU -- two gloo ranks shard the glaciers, each rank
produces a (loss, dtheta) for its shard and the single collective of the path sums them
(SIA2D_grad!: sum(losses) + aggregate_grad, gradient.jl:14,25)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import _odinn_import

    odinn = _odinn_import.load()
    r, w, _ = odinn.init_distributed("gloo")
    cells = [192 * 160, 96 * 80, 128 * 112, 160 * 128, 64 * 48]
    mine = odinn.shard_glaciers(cells, w)[r]
    # stand-in per-glacier results (the GPU parity tests pin the real ones)
    rng = [np.random.default_rng(100 + g) for g in range(len(cells))]
    per = [(float(x.random()), x.standard_normal(83)) for x in rng]
    loss = sum(per[g][0] for g in mine)
    dth = sum((per[g][1] for g in mine), np.zeros(83))
    L, D = odinn.allreduce_loss_grad(loss, dth)
    q.put((r, mine, L, D))
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_matches_serial_sum():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    rng = [np.random.default_rng(100 + g) for g in range(5)]
    per = [(float(x.random()), x.standard_normal(83)) for x in rng]
    Ltot = sum(p[0] for p in per)
    Dtot = sum((p[1] for p in per), np.zeros(83))
    shards = sorted(res)[0][1], sorted(res)[1][1]
    assert sorted(shards[0] + shards[1]) == list(range(5)) and shards[0] and shards[1]
    for r, mine, L, D in res:
        assert abs(L - Ltot) < 1e-12 and np.allclose(D, Dtot, rtol=1e-13, atol=1e-13)
    # both ranks hold identical reduced values (determinism of the 2-rank sum)
    assert res[0][2] == res[1][2] and np.array_equal(res[0][3], res[1][3])
