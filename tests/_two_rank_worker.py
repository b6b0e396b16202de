"""Worker of tests/test_gpu_two_rank.py: SIA2D_grad_b on 4 ragged glaciers with a per-glacier classical law
(GlacierWideInv) and a trainable initial condition, as ONE rank or as rank r of 2 (torch.distributed.run; both ranks on
device 0, gloo collectives -- ODINN_DEVICE=0 ODINN_DIST_BACKEND=gloo).  Writes loss and gradient to argv[1]."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _odinn_import  # noqa: E402

odinn = _odinn_import.load()


def alpine(nx, ny, dx=50.0, hmax=160.0, slope=0.1):  # same formula as the oracle's synthetic_alpine (tests only)
    x = (np.arange(nx) * dx)[:, None]
    y = (np.arange(ny) * dx)[None, :]
    yc = ny * dx / 2
    B = 2200.0 - slope * x + 300.0 * ((y - yc) / yc) ** 2
    ell = ((x - 0.45 * nx * dx) / (0.38 * nx * dx)) ** 2 + ((y - yc) / (0.30 * ny * dx)) ** 2
    return np.asfortranarray(np.maximum(0.0, hmax * (1.0 - ell))), np.asfortranarray(B + 0.0 * ell)


def main(out):
    k, step = 5, 1.0 / 96.0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    p = odinn.Parameters(simulation=odinn.SimulationParameters(tspan=(2010.0, 2010.0 + (k - 1) * step), multiprocessing=world > 1),
                         solver=odinn.SolverParameters(reltol=1e-10, step=step),
                         hyper=odinn.Hyperparameters(optimizer=odinn.LBFGS(), epochs=5))
    p.UDE.grad = odinn.DiscreteAdjoint()
    p.UDE.empirical_loss_function = odinn.MultiLoss(
        losses=(odinn.LossH(), odinn.InitialThicknessRegularization(t0=2010.0)), lambdas=(1.5, 2e-3))
    shapes = [(70, 57), (48, 40), (96, 64), (55, 47)]
    gl = []
    rng = np.random.default_rng(3)
    ts = [2010.0 + j * step for j in range(k)]
    for i, (nx, ny) in enumerate(shapes):
        H0, B = alpine(nx, ny)
        g = odinn.Glacier2D(f"SYN-{i}", H0, B, 50.0, 50.0, A=2e-17 * (1 + 0.3 * i))
        g.thicknessData = odinn.ThicknessData(ts, [H0 * (1.0 - 0.02 * j) + 0.3 * rng.random((nx, ny)) * (H0 > 0) for j in range(k)])
        gl.append(g)
    reg = odinn.GlacierWideInv(p, gl, "A")
    ic = odinn.InitialCondition(p, gl)
    model = odinn.Model(odinn.SIA2Dmodel(p, A=odinn.LawA(p, scalar=True)), regressors={"A": reg, "IC": ic})
    inv = odinn.Inversion(model, gl, p)
    th = model.theta.copy()
    th[:reg.theta.size] += 0.1 * np.arange(reg.theta.size)
    th[reg.theta.size:] *= 1.0 + 0.02 * np.random.default_rng(5).standard_normal(th.size - reg.theta.size)
    dth = np.zeros_like(th)
    loss = odinn.SIA2D_grad_b(dth, th, inv)
    rank = int(os.environ.get("RANK", "0"))
    np.savez(f"{out}.rank{rank}.npz", loss=loss, dth=dth, mine=np.array(inv._mine), world=world)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
