"""Worker of tests/test_gpu_rccl_ranks.py: rank r of N real RCCL ranks, ONE GPU PER RANK (device = LOCAL_RANK), through the
library's own communicator (odinn_comm_init_rank) and odinn_batch_loss_grad == SIA2D_grad! (gradient.jl:6-31).
argv: out-prefix, case ("ok" | "fail")."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _odinn_import  # noqa: E402

odinn = _odinn_import.load()
SHAPES = [(70, 57), (48, 40), (96, 64), (55, 47)]
TS = [2010.0 + j / 96.0 for j in range(5)]


def alpine(nx, ny, dx=50.0, hmax=160.0, slope=0.1):
    x = (np.arange(nx) * dx)[:, None]
    y = (np.arange(ny) * dx)[None, :]
    yc = ny * dx / 2
    B = 2200.0 - slope * x + 300.0 * ((y - yc) / yc) ** 2
    ell = ((x - 0.45 * nx * dx) / (0.38 * nx * dx)) ** 2 + ((y - yc) / (0.30 * ny * dx)) ** 2
    return np.asfortranarray(np.maximum(0.0, hmax * (1.0 - ell))), np.asfortranarray(B + 0.0 * ell)


def build(mine, device):
    """this rank's shard: glaciers `mine` of the four, default A(T) law (theta replicated), thickness data at every stop"""
    nn = odinn.NeuralNetwork(odinn.Parameters(), seed=7)
    ph = odinn.PhysicalParameters()
    mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA)
    b = odinn.GlacierBatch([SHAPES[i] for i in mine], [50.0] * len(mine), T=[-6.0 - i for i in mine], device=device)
    for k, i in enumerate(mine):
        H0, B = alpine(*SHAPES[i])
        b.set_fields(k, H0, B)
        b.set_reference(k, TS, [H0 * (1.0 - 0.02 * j) for j in range(len(TS))], 3)
    b.set_law(odinn.LAW_NN_A_SCALAR, mlp, nn.theta)
    return b, nn.theta


def main(out, case):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:  # the single-rank reference result
        b, th = build([0, 1, 2, 3], 0)
        Ld, gd = b.batch_loss_grad(None, TS, theta=th, reltol=1e-10)
        Lc, gc = b.batch_loss_grad(None, TS, theta=th, continuous=True, n_quadrature=12, reltol=1e-10)
        np.savez(f"{out}.rank0.npz", Ld=Ld, gd=gd, Lc=Lc, gc=gc)
        return
    r, w, local = odinn.init_distributed("nccl")  # torch's group only carries the 128-byte unique id; attaches the library's communicator
    comm = odinn.api._DIST["comm"]
    assert comm is not None and comm.rank_size() == (rank, world)
    mine = [i for i in range(4) if i % world == rank]
    b, th = build(mine, local)
    if case == "ok":
        Ld, gd = b.batch_loss_grad(comm, TS, theta=th, reltol=1e-10)
        Lc, gc = b.batch_loss_grad(comm, TS, theta=th, continuous=True, n_quadrature=12, reltol=1e-10)
        x = comm.allreduce_sum(np.arange(5, dtype=float) + rank)
        np.savez(f"{out}.rank{rank}.npz", Ld=Ld, gd=gd, Lc=Lc, gc=gc, x=x, mine=np.array(mine))
    else:
        # a rank-LOCAL failure (maxiters on rank 1 only): the status slot of the all-reduce makes EVERY rank return an error
        # instead of rank 0 blocking in ncclAllReduce forever
        err = ""
        try:
            b.batch_loss_grad(comm, TS, theta=th, reltol=1e-10, maxiters=(3 if rank == 1 else 10 ** 6))
        except odinn.OdinnError as e:
            err = str(e)
        # ... and the communicator is still usable afterwards
        L2, g2 = b.batch_loss_grad(comm, TS, theta=th, reltol=1e-10)
        np.savez(f"{out}.rank{rank}.npz", err=np.array(err), L2=L2, g2=g2)
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
