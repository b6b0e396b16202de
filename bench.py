#!/usr/bin/env python
"""bench.py -- headline benchmark of the SIA2D(+NN_theta) hot path on MI355X.

Workload (BASELINE.json configs[4], the config the scaling metric is quoted on, and it fits one GPU): the JOB is 64
synthetic 1024x1024 fp64 ice caps (per-glacier random radius, bed phase, seed 1234 + global glacier index), sharded over
the N GPUs of the run -- all 64 on one GPU at N = 1, 8 per GPU at N = 8 (strong scaling: "glacier-batch throughput at 8
GPUs vs 1") -- resident in HBM, with the "CuffeyPaterson-style" law of configs[2]: A = NN_theta(T) -- a 2-hidden-layer x
16-unit MLP on a gridded long-term temperature, hoisted into a dual-grid A field once per theta exactly as the reference
evaluates LawA (src/laws/Laws.jl:339-358).  The fixed 8-glaciers-per-GPU (weak) figure is in aux.weak_8_per_gpu.

A "step" is one pass of the hot path over that batch exactly as odinn_solve launches it: one
RDPK3Sp35 time step (5 RHS + stage updates per cell) + the controller (error-norm reduction, PID);
a stop's snapshot is stored by the next step launch from the state it loads (batches with a mass
balance launch a post-step kernel as well).  One cell-step = one cell through one fused RHS + stage update, so a step
is 5 * cells cell-steps.  The hoisted law is evaluated once per theta (per solve, >= 100 steps), not
per step: its cost is reported next to the step (aux.law_field_ms) and inside the end-to-end figure
aux.full_solve (a real adaptive odinn_solve: law evaluation, initial step size, every step, snapshots,
host polls).  `value` is measured with the exact ice-free-tile shortcut OFF (dense work); what
odinn_solve runs by default (shortcut on) is in aux.

    python bench.py --gpus N --steps K --warmup W
    (N > 1 without a torch.distributed environment: re-executes itself under
     `python -m torch.distributed.run --nproc-per-node N`, one rank per GPU; fails if fewer than N
     devices are visible.  Glaciers shard with no data-path collective; the only collective of the path,
     ONE ncclAllReduce of [n_failed, loss, dtheta] inside libodinn_hip (odinn_batch_loss_grad), runs in the grad-eval leg.
     If the library's communicator cannot be created the run FAILS -- no silent fallback to another reduction.)

Prints ONE JSON line (rank 0):
  roofline            dominant kernel of the timed region, k_rk_fused_strip<gridded A, 8 rows>: a whole
                      RDPK3Sp35 step in one launch.  It moves ~32 B/cell (R u,B,A  W u'), so it is NOT
                      HBM-bound: bound = fp64 VALU.  achieved = useful fp64 flops per launch / launch time;
                      flops per cell-stage come from the committed PMC pass (profiles/r03/pmc_roofline.json:
                      SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 per launch / executed cell-stages incl. halo), useful
                      cell-stages = 5 * cells (halo recomputation is not counted as useful work).
  roofline_hbm        the north-star kernel: the fused SIA2D+NN RHS stencil (k_dhdt, gridded A) on a working
                      set PAST the 256 MiB Infinity Cache (32 x 1024^2: 1 GiB), 32 B/cell.
  roofline_per_stage  k_rk_stage<2> of the HBM-bound per-stage schedule, also past the Infinity Cache.
  grad_evals_per_s    forward solve + adjoint (+ all-reduce), discrete and continuous adjoint.
  cpu_baseline        the oracle's C restatement (oracle/sia2d_oracle.c, OpenMP) on the host cores.
"""
import argparse
import json
import os

# Kernel arguments in device memory: the HIP runtime's launch-latency setting for MI300-class parts (read when the
# runtime initialises, so before torch / the library make their first HIP call).  Already the default of ROCm 7.2 on
# gfx950 (no difference measured with or without this line); pinned here, for THIS process only, because the step loop is
# a chain of dependent launches and an explicit 0 costs 3 % per step at 8 x 1024^2 and 27 % on the 4-glacier gradients.
# A value set in the environment wins.  (The library and its Python layer do not touch the process environment.)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz)
B_PER_CELL_STAGE2 = 56.0  # interior stage: R u,B,tmp,utilde  W u',tmp,utilde
B_PER_CELL_STEP = 264.0  # 40 + 56 + 56 + 64 + 48 over the five stages
B_PER_CELL_DHDT = 24.0  # R H,B  W dH
B_PER_CELL_DHDT_NN = 32.0  # + the dual-grid A field
B_PER_CELL_VJPH = 32.0
B_PER_CELL_FUSED = 24.0  # fused step kernel: R u,B  W u'
B_PER_CELL_FUSED_NN = 32.0  # + the dual-grid A field
# fp64 flops per EXECUTED cell-stage of the strip kernel, fallback when profiles/r0x/pmc_roofline.json is absent:
# profiles/r01/pmc_fused_strip_sq.md: (9.60 + 14.03 + 2 x 11.12) M wave-instr x 64 lanes / (2888 tiles x 4096 cells x 5)
FLOP_PER_CELL_STAGE_FALLBACK = 49.7
PMC_FILES = [os.path.join(ROOT, "profiles", r, "pmc_roofline.json") for r in ("r06", "r04", "r03", "r02")]  # newest committed pass first


def make_glacier(n, gidx, dx=100.0):
    """Config-5 ice cap: B = 500 + 50 sin cos (random phase) + 0.01 x; H0 = max(0, 800 (1-(r/R)^2))."""
    rng = np.random.default_rng(1234 + gidx)
    x = (np.arange(n) * dx)[:, None]
    y = (np.arange(n) * dx)[None, :]
    Lx = n * dx
    ph = rng.uniform(0, 2 * np.pi, 2)
    B = 500.0 + 50.0 * np.sin(2 * np.pi * x / Lx + ph[0]) * np.cos(2 * np.pi * y / Lx + ph[1]) + 0.01 * x
    R = rng.uniform(0.3, 0.42) * Lx
    r = np.sqrt((x - Lx / 2) ** 2 + (y - Lx / 2) ** 2)
    H0 = np.maximum(0.0, 800.0 * (1.0 - (r / R) ** 2))
    A = 10.0 ** rng.uniform(np.log10(1e-18), np.log10(4e-17))
    return np.asfortranarray(H0), np.asfortranarray(B + 0.0 * H0), A


def temperature_field(H0, B):
    """configs[2](i): T = -5 - 6.5e-3 (S - mean S) deg C on the dual grid."""
    S = B + H0
    Sd = 0.25 * (S[:-1, :-1] + S[1:, :-1] + S[:-1, 1:] + S[1:, 1:])
    return np.asfortranarray(-5.0 - 6.5e-3 * (Sd - S.mean()))


def alpine(nx, ny, dx=50.0, hmax=110.0, slope=0.08):
    """configs[3] stand-in (the README glaciers are not in the image): gentle valley glacier."""
    x = (np.arange(nx) * dx)[:, None]
    y = (np.arange(ny) * dx)[None, :]
    yc = ny * dx / 2
    B = 2200.0 - slope * x + 300.0 * ((y - yc) / yc) ** 2
    ell = ((x - 0.45 * nx * dx) / (0.38 * nx * dx)) ** 2 + ((y - yc) / (0.30 * ny * dx)) ** 2
    return np.asfortranarray(np.maximum(0.0, hmax * (1.0 - ell))), np.asfortranarray(B + 0.0 * ell)


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` with no torch.distributed environment: launch the N ranks ourselves."""
    import torch

    have = torch.cuda.device_count()
    if have < n and "ODINN_BENCH_DEVICE" not in os.environ:  # (ODINN_BENCH_DEVICE: dry run of the N-rank path on one device)
        raise SystemExit(f"bench.py --gpus {n}: only {have} HIP device(s) visible -- refusing to report a {n}-GPU number")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--glaciers", type=int, default=64, help="glaciers of the whole job (BASELINE configs[4]: 64), sharded over the GPUs")
    ap.add_argument("--glaciers-per-gpu", type=int, default=0, help="> 0: fix the per-GPU share instead (weak scaling; the job is then N x this)")
    ap.add_argument("--hbm-glaciers", type=int, default=32, help="batch for the HBM-bound per-kernel figures (past the Infinity Cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grad-eval", action="store_true")
    ap.add_argument("--no-hbm-sweep", action="store_true")
    ap.add_argument("--no-full-config", action="store_true", help="(accepted for old command lines; the 64-glacier job IS the timed workload now)")
    ap.add_argument("--no-weak", action="store_true", help="skip aux.weak_8_per_gpu (the fixed 8-glaciers-per-GPU figure)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--headline-only", action="store_true", help="only the timed headline loop (for `rocprofv3 --kernel-trace --stats`: the "
                    "kernel averages are then those of THIS workload, not mixed with the auxiliary batches)")
    ap.add_argument("--repeats", type=int, default=7, help="repeats of the timed region inside this invocation (value = the median)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args.gpus)  # does not return

    # stdout carries ONE JSON line and nothing else: libraries that print banners on file descriptor 1 (RCCL's version block at
    # communicator creation, gloo's rank messages) are sent to stderr for the duration of the run
    sys.stdout.flush()
    _stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    import _odinn_import

    odinn = _odinn_import.load()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    dist = None
    # ODINN_BENCH_BACKEND=gloo ODINN_BENCH_DEVICE=0: dry run of the N > 1 path on a ONE-GPU box (every rank
    # on device 0, collectives over gloo); the driver's runs use the defaults: RCCL, one GPU per rank
    backend = os.environ.get("ODINN_BENCH_BACKEND", "nccl")
    if "ODINN_BENCH_DEVICE" in os.environ:
        local = int(os.environ["ODINN_BENCH_DEVICE"])
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} HIP device(s) visible")
    red_dev = f"cuda:{local}" if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        odinn.api._DIST.update(init=True, rank=rank, world=world, local=local, device=red_dev)
        comm_note = "torch.distributed (" + backend + "): explicit dry-run backend, NOT the library's communicator"
        if backend == "nccl":  # the all-reduce of [loss, dtheta] runs inside libodinn_hip (odinn_comm_*, RCCL over xGMI)
            err = ""
            try:
                odinn.api.attach_rccl_comm(local)
            except Exception as e:
                odinn.api._DIST["comm"] = None
                err = str(e)[:300]
            ok = torch.tensor([1.0 if odinn.api._DIST.get("comm") is not None else 0.0], device=red_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # every rank learns whether ANY rank failed
            if float(ok.item()) == 0.0:
                # never credit a scaling line to a path that is not the library's: no fallback to torch's group
                raise SystemExit(f"bench.py: odinn_comm_init_rank failed on rank {rank if err else '(another)'}: {err} -- refusing to "
                                 "run the N-GPU bench with a different reduction")
            comm_rank, comm_n = odinn.api._DIST["comm"].rank_size()
            if comm_n != world:
                raise SystemExit(f"bench.py: the library's communicator has {comm_n} ranks, WORLD_SIZE is {world}")
            comm_note = f"libodinn_hip odinn_comm_* (ncclAllReduce inside the library, communicator of {comm_n} ranks)"
    if odinn.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")

    T = odinn._lib
    ph = odinn.PhysicalParameters()
    n = args.size
    if args.glaciers_per_gpu > 0:  # weak scaling on request: fixed per-GPU share
        G, first, G_job, scaling = args.glaciers_per_gpu, rank * args.glaciers_per_gpu, world * args.glaciers_per_gpu, "weak"
    else:  # the configs[4] job, sharded: rank r owns a contiguous block (the first G_job % world ranks hold one more)
        G_job, scaling = args.glaciers, "strong"
        if G_job < world:
            raise SystemExit(f"bench.py: {G_job} glaciers cannot be sharded over {world} GPUs")
        G = G_job // world + (1 if rank < G_job % world else 0)
        first = rank * (G_job // world) + min(rank, G_job % world)
    gl = [make_glacier(n, first + k) for k in range(G)]

    def make_batch(glist):
        bb_ = odinn.GlacierBatch([(n, n)] * len(glist), [100.0] * len(glist), A=[g[2] for g in glist], device=local)
        for k, (H0, B, A) in enumerate(glist):
            bb_.set_fields(k, H0, B)
            bb_.set_T_field(k, temperature_field(H0, B))
        return bb_

    b = make_batch(gl)
    cells = b.cells
    cells_job = G_job * n * n
    # configs[2]: "2-layer/16-unit NN_theta law (CuffeyPaterson-style)": A = minA + (maxA - minA) MLP(T)
    mlpA = odinn.MLPSpec([1, 16, 16, 1], [odinn.ACT_SOFTPLUS, odinn.ACT_SOFTPLUS, odinn.ACT_SIGMOID],
                         None, odinn.POST_AFFINE, ph.minA, ph.maxA)
    thetaA = np.random.default_rng(1234).uniform(-0.5, 0.5, mlpA.n_params)
    b.set_law(odinn.LAW_NN_A_GRIDDED, mlpA, thetaA)

    sched_in_effect = b.get_schedule()

    def barrier():
        b.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- timed region: K steps of the real per-step launch sequence ----------------------------------
    # The region (barrier + synchronize, EXACTLY K steps, synchronize, max over ranks) is repeated args.repeats times inside
    # this invocation and `value` is the MEDIAN repeat: a 20-step driver run times ~17 ms, where a single bracket moves by
    # several per cent with whatever else the host does at that moment.  `steps` stays the per-repeat count; every repeat is
    # in aux.timed_region_repeats.
    # The dominant kernel of every timed step sits between two HIP events on the library's stream (odinn_bench_kernel_events), read
    # after the bracket closes: roofline.ms_per_launch is the kernel's time in the SAME launches that `ms_per_step` times.
    b.bench_prepare()
    b.bench_kernel_events(True)
    b.bench_enqueue(T.TIMED_SOLVE_STEP, 0, args.warmup)
    b.bench_kernel_ms()  # (drops the warm-up's pairs)
    repeats, kernel_ms = [], []
    done = args.warmup
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        b.bench_enqueue(T.TIMED_SOLVE_STEP, done, args.steps)
        b.sync()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        kms, kn_l = b.bench_kernel_ms()
        kernel_ms.append(kms / max(kn_l, 1))
        done += args.steps
        el = t1 - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
            dist.barrier()
        repeats.append(el)
    elapsed = float(np.median(repeats))
    i_med = int(np.argsort(repeats)[len(repeats) // 2])   # the median repeat: its own kernel time goes with its step time
    ms_kernel_in_loop = kernel_ms[i_med]
    b.bench_kernel_events(False)
    cellsteps = 5.0 * cells_job * args.steps  # the whole job: every rank's glaciers
    value = cellsteps / elapsed

    if args.headline_only:
        if rank == 0:
            print(json.dumps({"metric": "cell-steps/s (forward SIA2D+NN)", "value": value, "unit": "cell-steps/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "dtype": "f64",
                              "fused_kernel_ms_per_launch_in_loop": ms_kernel_in_loop, "repeats_ms_per_step": [e / args.steps * 1e3 for e in repeats],
                              "config": {"workload": f"{G} x {n}^2 glaciers per GPU, headline loop only (--headline-only)"}}), flush=True)
        b.close()
        return

    # ---- the fixed per-GPU share (8 glaciers per GPU, weak scaling): same launches, same bracket -------------------
    weak = None
    if not args.no_weak and args.glaciers_per_gpu <= 0:
        if G == 8:
            weak = {"value": value, "ms_per_step": elapsed / args.steps * 1e3, "note": "the timed job already has 8 glaciers per GPU"}
        else:
            b8 = make_batch(gl[:8] if G >= 8 else [make_glacier(n, first + k) for k in range(8)])
            b8.set_law(odinn.LAW_NN_A_GRIDDED, mlpA, thetaA)
            b8.bench_prepare()
            b8.bench_enqueue(T.TIMED_SOLVE_STEP, 0, args.warmup)
            b8.sync(); barrier()
            tw0 = time.perf_counter()
            b8.bench_enqueue(T.TIMED_SOLVE_STEP, args.warmup, args.steps)
            b8.sync()
            tw = time.perf_counter() - tw0
            if dist is not None:
                tt = torch.tensor([tw], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tw = float(tt.item())
            weak = {"value": 5.0 * b8.cells * world * args.steps / tw, "ms_per_step": tw / args.steps * 1e3,
                    "fused_step_kernel_ms": b8.time_kernel(T.TIMED_FUSED_STEP, iters=30, warmup=5),
                    "dhdt_nn_gridded_in_cache_GBs": B_PER_CELL_DHDT_NN * b8.cells / (b8.time_kernel(T.TIMED_DHDT, iters=50, warmup=5) * 1e-3) / 1e9,
                    "note": f"8 x {n}^2 glaciers per GPU on each of the {world} GPU(s) (the round-1..3 headline workload; working sets of "
                            "192-448 MiB sit partly in the 256 MiB Infinity Cache): whole-job cell-steps/s, max over ranks"}
            b8.close()
            del b8

    # ---- the same launches bracketed by HIP events on the library's own stream ----------------------
    ev = lambda which, iters=30, warmup=5: b.time_kernel(which, iters=iters, warmup=warmup)
    aux = {}
    aux["timed_region_repeats"] = {
        "n": len(repeats), "steps_each": args.steps, "ms_per_step": [e / args.steps * 1e3 for e in repeats],
        "fused_kernel_ms_per_launch": kernel_ms,
        "value_min": cellsteps / max(repeats), "value_median": value, "value_max": cellsteps / min(repeats),
        "note": "`value` / `ms_per_step` of the line are the median repeat; every repeat is a barrier + synchronize bracket around "
                "exactly `steps` steps, max over ranks"}
    if world > 1:
        aux["loss_grad_allreduce"] = comm_note
    ms_step_nn = ev(T.TIMED_SOLVE_STEP, args.steps, args.warmup)
    ms_fused_nn = ev(T.TIMED_FUSED_STEP)
    ms_fused_nn_skip = ev(T.TIMED_FUSED_STEP_SKIP)
    ms_law = ev(T.TIMED_LAW_FIELD, 10, 2)
    ms_dhdt_nn = ev(T.TIMED_DHDT, 50, 5)
    aux.update({
        "solve_step_ms_hip_events": ms_step_nn,
        "law_field_ms": ms_law,
        "law_field_note": "k_law_field: the 2x16 MLP evaluated on every dual node, once per theta (= once per solve; outside the timed steps, inside full_solve)",
        "fused_step_with_ice_free_shortcut_ms": ms_fused_nn_skip,
        "fused_step_with_ice_free_shortcut_cellsteps_per_s": 5.0 * cells / (ms_fused_nn_skip * 1e-3),
        "per_rank_note": "HIP-event figures in aux and the roofline blocks are measured on rank 0's shard (its own stream)",
        "ice_free_shortcut_note": "odinn_solve's default: workgroups whose halo region has u == 0 skip the stages "
                                  "(bit-identical); `value` is measured with the shortcut OFF (dense work)",
    })

    # ---- end to end: a real adaptive solve of the same batch with the NN law (half a year, monthly stops) ----------
    try:
        ts_f = [2010.0 + k / 12.0 for k in range(7)]
        b.solve(ts_f, reltol=1e-6, dense=1)  # warm
        barrier()
        tf0 = time.perf_counter()
        st_f = b.solve(ts_f, reltol=1e-6, dense=1)
        b.sync()
        tf = time.perf_counter() - tf0
        nst = [s_.naccept + s_.nreject for s_ in st_f]
        aux["full_solve"] = {
            "ms": tf * 1e3, "steps_per_glacier": nst if len(nst) <= 8 else [min(nst), max(nst)], "cellsteps_per_s": 5.0 * (n * n) * sum(nst) / tf,
            "us_per_step_of_the_slowest_glacier": tf * 1e6 / max(nst),
            "note": "odinn_solve(6 monthly stops, reltol 1e-6, dense work) with A = NN_theta(T): hoisted-law evaluation, Hairer-Wanner "
                    "initial step, every accepted and rejected RDPK3Sp35 step, snapshots and host polls included; glaciers step "
                    "independently, so the count is the sum over glaciers",
        }
    except Exception as e:
        aux["full_solve_error"] = str(e)[:200]

    # ---- constant A (BASELINE configs[1]) on the same batch ------------------------------------------
    b.set_law(odinn.LAW_CONST_A)
    ms_step_c = ev(T.TIMED_SOLVE_STEP, args.steps, args.warmup)
    ms_fused_c = ev(T.TIMED_FUSED_STEP)
    ms_fused_c_skip = ev(T.TIMED_FUSED_STEP_SKIP)
    ms_solve_staged = ev(T.TIMED_SOLVE_STEP_STAGED, 20, 3)
    ms_cfl = ev(T.TIMED_EULER_CFL, 50, 5)
    ms_adjf = ev(T.TIMED_ADJ_FUSED_STEP, 20, 3)
    ms_adj = ev(T.TIMED_ADJ_STAGE2, 20, 3)
    aux.update({
        "constA_ms_per_step": ms_step_c,
        "constA_cellsteps_per_s": 5.0 * cells / (ms_step_c * 1e-3),
        "constA_fused_step_kernel_ms": ms_fused_c,
        "constA_fused_step_with_ice_free_shortcut_ms": ms_fused_c_skip,
        "constA_note": "BASELINE configs[1]: constant A per glacier, same grids, same launch sequence (HIP events)",
        "per_stage_schedule_ms_per_step": ms_solve_staged,
        "per_stage_schedule_cellsteps_per_s": 5.0 * cells / (ms_solve_staged * 1e-3),
        "euler_cfl_step_ms": ms_cfl,
        "euler_cfl_cellsteps_per_s": cells / (ms_cfl * 1e-3),
        "euler_cfl_note": "explicit Euler with CFL-limited dt (scheme 3): ONE cell-step per cell per launch, 24 B per cell-step",
        "adj_fused_step_ms": ms_adjf,
        "adj_stage2_ms": ms_adj,
        "adj_fused_step_note": "k_adj_fused_strip: a whole RDPK3Sp35 step of the reverse ODE of the continuous adjoint in one "
                               "kernel (R lam,H_j,H_j+1,B  W lam' = 40 B/cell) against 5 x adj_stage2_ms for the staged schedule",
        "adj_fused_step_speedup_vs_5_stage_kernels": 5.0 * ms_adj / ms_adjf,
    })

    # ---- HBM-bound kernels on a working set past the 256 MiB Infinity Cache -------------------------
    # (rank 0's shard itself when it holds >= 32 glaciers -- the 64-glacier job at N = 1 --, otherwise a 32-glacier batch)
    hbm = {}
    if rank == 0 and not args.no_hbm_sweep:
        try:
            own = G >= args.hbm_glaciers
            Gb = G if own else args.hbm_glaciers
            bb = b if own else make_batch([gl[k] if k < G else make_glacier(n, 1000 + k) for k in range(Gb)])
            bb.set_law(odinn.LAW_CONST_A)
            cb = bb.cells
            evb = lambda which, iters=20, warmup=3: bb.time_kernel(which, iters=iters, warmup=warmup)
            rows = {}
            for name, which, bpc in (("dhdt", T.TIMED_DHDT, B_PER_CELL_DHDT), ("rk_stage2", T.TIMED_RK_STAGE2, B_PER_CELL_STAGE2),
                                     ("euler_cfl", T.TIMED_EULER_CFL, 24.0), ("vjp_H", T.TIMED_VJP_H, B_PER_CELL_VJPH),
                                     ("vjp_theta", T.TIMED_VJP_THETA, 24.0), ("adj_stage2", T.TIMED_ADJ_STAGE2, 72.0)):
                ms = evb(which)
                rows[name] = {"ms": ms, "bytes_per_cell": bpc, "GBs": bpc * cb / (ms * 1e-3) / 1e9,
                              "frac_of_hbm_peak": bpc * cb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            ms_fused_big = evb(T.TIMED_FUSED_STEP)
            rows["fused_step_constA"] = {"ms": ms_fused_big, "cellsteps_per_s": 5.0 * cb / (ms_fused_big * 1e-3)}
            bb.set_law(odinn.LAW_NN_A_GRIDDED, mlpA, thetaA)
            ms = evb(T.TIMED_DHDT)
            rows["dhdt_nn_gridded"] = {"ms": ms, "bytes_per_cell": B_PER_CELL_DHDT_NN, "GBs": B_PER_CELL_DHDT_NN * cb / (ms * 1e-3) / 1e9,
                                       "frac_of_hbm_peak": B_PER_CELL_DHDT_NN * cb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            ms_fused_big_nn = evb(T.TIMED_FUSED_STEP)
            ms_step_big_nn = evb(T.TIMED_SOLVE_STEP)
            rows["fused_step_nn_gridded"] = {"ms": ms_fused_big_nn, "cellsteps_per_s": 5.0 * cb / (ms_fused_big_nn * 1e-3)}
            rows["solve_step_nn_gridded"] = {"ms": ms_step_big_nn, "cellsteps_per_s": 5.0 * cb / (ms_step_big_nn * 1e-3)}
            hbm = {"glaciers": Gb, "cells": cb, "working_set_note": f"{Gb} x {n}^2 fp64: {8 * cb / 2**20:.0f} MiB per field "
                   "(every kernel's working set > 512 MiB, past the 256 MiB Infinity Cache)", "kernels": rows}
            if not own:
                bb.close()
                del bb
        except Exception as e:
            hbm = {"error": str(e)[:300]}
    aux["hbm_past_infinity_cache"] = hbm
    aux["weak_8_per_gpu"] = weak

    # ---- cross-checks: what a plain device copy / triad reaches on this box, and the
    #      PCIe-inclusive rate of the host-pointer seams (never part of `value`) ------------------
    try:
        dev = f"cuda:{local}"
        nb = 1 << 27  # 1 GiB per fp64 array
        xa = torch.full((nb,), 1.0, dtype=torch.float64, device=dev)
        xb = torch.full((nb,), 2.0, dtype=torch.float64, device=dev)
        xc = torch.empty_like(xa)

        def _ev(fn, iters=10):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters

        ms_copy = _ev(lambda: xc.copy_(xa))
        ms_triad = _ev(lambda: torch.add(xa, xb, alpha=3.0, out=xc))
        aux["hbm_copy_GBs_measured"] = 2 * 8.0 * nb / (ms_copy * 1e-3) / 1e9
        aux["hbm_triad_GBs_measured"] = 3 * 8.0 * nb / (ms_triad * 1e-3) / 1e9
        aux["hbm_crosscheck_note"] = ("torch device copy (R+W) / triad c = a + 3 b (2R+W) on 1 GiB fp64 arrays: the rate a "
                                      "trivially streaming kernel reaches on this box, next to the 8 TB/s datasheet peak")
        del xa, xb, xc
        torch.cuda.empty_cache()
        # host-pointer seam: H, B (and dH back) cross PCIe on every call
        Hh = gl[0][0]
        b.dhdt(0, Hh)
        tp0 = time.perf_counter()
        for _ in range(5):
            b.dhdt(0, Hh)
        ms_host = (time.perf_counter() - tp0) / 5 * 1e3
        tp0 = time.perf_counter()
        for k, (H0, B, A) in enumerate(gl):
            b.set_fields(k, H0, B)
        b.sync()
        ms_h2d = (time.perf_counter() - tp0) * 1e3
        tp0 = time.perf_counter()
        for k in range(G):
            b.H(k)
        ms_d2h = (time.perf_counter() - tp0) * 1e3
        ms_job = elapsed * 1e3  # (rank 0's share of the job: its G glaciers, the timed steps)
        aux["pcie_dhdt_host_pointers_ms_per_call"] = ms_host
        aux["pcie_dhdt_host_pointers_cells_per_s"] = n * n / (ms_host * 1e-3)
        aux["pcie_upload_H0_B_ms"] = ms_h2d
        aux["pcie_download_H_ms"] = ms_d2h
        aux["pcie_inclusive_cellsteps_per_s"] = 5.0 * cells * args.steps / ((ms_job + ms_h2d + ms_d2h) * 1e-3)
        aux["pcie_note"] = (f"inclusive = upload H0,B of all {G} glaciers + {args.steps} resident steps + download H; "
                            "pageable host memory; reported for reference, never part of `value`")
    except Exception as e:
        aux["crosscheck_error"] = str(e)[:200]

    # ---- grad-eval/s: forward solve + adjoint + all-reduce of [loss, dtheta] --------------------------
    grad = None
    if not args.no_grad_eval:
        grad = {}
        try:
            # (i) the bench workload: 8 x 1024^2 per GPU, default A(T) MLP (scalar T per glacier), 2 years of monthly
            #     thickness snapshots (k = 25), reltol 1e-8 -- the inversion settings of BASELINE configs[3]
            nn = odinn.NeuralNetwork(odinn.Parameters(), seed=666)
            mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA)
            b.set_law(odinn.LAW_NN_A_SCALAR, mlp, nn.theta)
            ts = [2010.0 + k / 12.0 for k in range(25)]
            for k in range(G):
                b.set_reference(k, ts, [gl[k][0] * (1.0 - 0.002 * j) for j in range(len(ts))], 3)

            comm = odinn.api._DIST.get("comm") if world > 1 else None  # the library's RCCL communicator (None: one rank)
            torch_reduce = world > 1 and comm is None  # only the explicit ODINN_BENCH_BACKEND=gloo dry run

            def timed(continuous, theta):
                # SIA2D_grad! == odinn_batch_loss_grad: this rank's forward solve + adjoint, then ONE ncclAllReduce of
                # [n_failed, loss, dtheta] inside the library; every rank returns the global loss and gradient
                def fn():
                    L_, g_ = b.batch_loss_grad(comm, ts, theta=theta, continuous=continuous, reltol=1e-8)
                    if torch_reduce:
                        L_, g_ = odinn.allreduce_loss_grad(L_, g_)
                    return L_, g_

                fn()  # warm
                barrier()
                tg0 = time.perf_counter()
                fn()
                b.sync()
                tg_ = time.perf_counter() - tg0
                if dist is not None:
                    tt_ = torch.tensor([tg_], dtype=torch.float64, device=red_dev)
                    dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                    tg_ = float(tt_.item())
                return tg_

            tg = timed(False, nn.theta)
            st = b.last_stats
            tgc = timed(True, nn.theta)
            rev = b.last_stats_rev[0] if getattr(b, "last_stats_rev", None) else None
            grad["bench_workload"] = {
                "discrete_adjoint": G_job / tg,
                "continuous_adjoint": G_job / tgc,
                "allreduce": comm_note if world > 1 else "none (one rank: odinn_batch_loss_grad with comm = NULL)",
                "comm_nranks": (comm.rank_size()[1] if comm is not None else 1),
                "sample": f"the {G_job} x {n}^2 glaciers of the job ({G} on this rank), default A(T) MLP ({len(nn.theta)} params), k = 25 monthly "
                          f"thickness snapshots over 2 yr, reltol 1e-8: {st[0].naccept}+{st[0].nreject} forward RK steps, 24 reverse-Euler "
                          f"VJP pairs (DiscreteAdjoint); ContinuousAdjoint defaults (reltol = abstol = 1e-8, dtmax = 1/12, 200 "
                          f"Gauss-Legendre nodes)" + (f": {rev.naccept}+{rev.nreject} reverse RK steps" if rev else "")
                          + "; one odinn_batch_loss_grad per evaluation (forward solve + adjoint + the all-reduce), max over ranks",
                "ms_per_grad_eval_batch_discrete": tg * 1e3,
                "ms_per_grad_eval_batch_continuous": tgc * 1e3,
            }
            # (i') the same with the headline's law: A = NN_theta(T) on the dual grid (2x16 MLP, 321 parameters; dtheta through
            #      the dual-grid accumulator and the wave-reduced backprop kernel)
            b.set_law(odinn.LAW_NN_A_GRIDDED, mlpA, thetaA)
            tgg = timed(False, thetaA)
            tggc = timed(True, thetaA)
            grad["bench_workload_gridded_law"] = {
                "discrete_adjoint": G_job / tgg,
                "continuous_adjoint": G_job / tggc,
                "sample": f"as bench_workload, but A = NN_theta(T) gridded with the 2x16 MLP ({len(thetaA)} params) of the headline",
            }
            # (i'') the per-node law Y = NN_theta(T_glacier, Hbar) (target :D_hybrid, default 2-3-10-3-1 net, the reference's default
            #       `:Linear` gradient interpolation) through its per-glacier table -- the library's default; the network itself in the
            #       stencil (law_table = 0) only with the cheaper DiscreteAdjoint
            try:
                mY = odinn.MLPSpec([2, 3, 10, 3, 1], [odinn.ACT_SOFTPLUS] * 3 + [odinn.ACT_SIGMOID], [(-25.0, 0.0), (0.0, 500.0)],
                                   odinn.POST_EXPMAX, 0.0, ph.maxA)
                thY = np.random.default_rng(1234).uniform(-0.5, 0.5, mY.n_params)
                b.set_law(odinn.LAW_NN_Y, mY, thY)
                tyd = timed(False, thY)
                tyc = timed(True, thY)
                revy = b.last_stats_rev[0] if getattr(b, "last_stats_rev", None) else None
                b.set_schedule(law_table=0)
                tynd = timed(False, thY)
                b.set_schedule()
                grad["bench_workload_Y_law"] = {
                    "discrete_adjoint": G_job / tyd, "continuous_adjoint": G_job / tyc, "discrete_adjoint_network_in_stencil": G_job / tynd,
                    "table_usable": b.law_table()["usable"],
                    "sample": "as bench_workload, but the Y law (86 parameters) with its table Y(Hbar) per glacier; ContinuousAdjoint: one fused "
                              "reverse launch per step" + (f", {revy.naccept}+{revy.nreject} reverse RK steps" if revy else "") +
                              ", the `:Linear` contractions of the quadrature nodes on lane streams",
                }
            except Exception as e:
                grad["bench_workload_Y_law"] = {"error": str(e)[:300]}
                b.set_schedule()
            # (i''') the per-node law U = NN_theta(Hbar, |grad S|) (target :D, default 2-3-10-3-1 net, inputs scaled as in the reference's
            #        tests, the reference's default `:None` gradient interpolation = exact backprop at every node) through the batch's
            #        bivariate table, whose resolution the library picks by measurement
            try:
                mU = odinn.MLPSpec([2, 3, 10, 3, 1], [odinn.ACT_SOFTPLUS] * 3 + [odinn.ACT_SIGMOID], [(0.0, 300.0), (0.0, 0.5)],
                                   odinn.POST_EXPMAX, 0.0, 50.0)
                thU = np.random.default_rng(1234).uniform(-0.5, 0.5, mU.n_params)
                b.set_law(odinn.LAW_NN_U, mU, thU)
                tud = timed(False, thU)
                tuc = timed(True, thU)
                revu = b.last_stats_rev[0] if getattr(b, "last_stats_rev", None) else None
                info = b.law_table()
                grad["bench_workload_U_law"] = {
                    "discrete_adjoint": G_job / tud, "continuous_adjoint": G_job / tuc,
                    "table_usable": info["usable"], "table_patches": info["n_intervals"], "table_max_rel_dev_from_network": info["max_rel_dev"],
                    "sample": "as bench_workload, but the U law (86 parameters, prescale (0, 300) x (0, 0.5), U <= 50 m/yr) through its table; "
                              "ContinuousAdjoint: one fused LDS-tile reverse launch per step" +
                              (f", {revu.naccept}+{revu.nreject} reverse RK steps" if revu else "") + ", theta-VJP by backprop at every node",
                }
            except Exception as e:
                grad["bench_workload_U_law"] = {"error": str(e)[:300]}
            b.set_law(odinn.LAW_CONST_A)
            # (ii) BASELINE configs[3]: 4 alpine glaciers (synthetic stand-ins of the README set), and the same set
            #      replicated to fill the GPU (the reference maps one glacier per worker process)
            if rank == 0:
                shapes4 = [(96, 80), (128, 112), (160, 128), (192, 160)]
                for Ga in (4, 512):
                    shapes = [shapes4[k % 4] for k in range(Ga)]
                    ba = odinn.GlacierBatch(shapes, [50.0] * Ga, T=[-9.0 + 0.5 * (k % 7) for k in range(Ga)], device=local)
                    cache = {s: alpine(*s) for s in shapes4}
                    for k, s in enumerate(shapes):
                        ba.set_fields(k, *cache[s])
                    nn2 = odinn.NeuralNetwork(odinn.Parameters(), seed=42)
                    ba.set_law(odinn.LAW_NN_A_SCALAR, mlp, nn2.theta)
                    ba.solve(ts, reltol=1e-8)
                    refs = [[ba.snapshot(k, j) for j in range(len(ts))] for k in range(4)]
                    for k in range(Ga):
                        ba.set_reference(k, ts, refs[k % 4], 3)
                    th0 = odinn.NeuralNetwork(odinn.Parameters(), seed=1234).theta
                    ba.loss_grad(ts, theta=th0, reltol=1e-8)
                    tq0 = time.perf_counter()
                    for _ in range(3):
                        ba.loss_grad(ts, theta=th0, reltol=1e-8)
                    td = (time.perf_counter() - tq0) / 3
                    steps_a = max(s.naccept + s.nreject for s in ba.last_stats)
                    ba.loss_grad_continuous(ts, theta=th0, reltol=1e-8)
                    tq0 = time.perf_counter()
                    ba.loss_grad_continuous(ts, theta=th0, reltol=1e-8)
                    tc = time.perf_counter() - tq0
                    grad[f"configs3_alpine_G{Ga}"] = {
                        "discrete_adjoint": Ga / td, "continuous_adjoint": Ga / tc,
                        "sample": f"{Ga} alpine glaciers (96x80 ... 192x160 cycling), default A(T) MLP, k = 25 monthly snapshots, "
                                  f"reltol 1e-8, <= {steps_a} forward RK steps; 1 GPU",
                    }
                    ba.close()
        except Exception as e:  # never lose the headline line to the extras
            grad["error"] = str(e)[:300]
        # BASELINE configs[2](ii): the 2x16 MLP as a Y law inlined per dual node in the stencil (fp64-compute-bound)
        try:
            mlp16 = odinn.MLPSpec([2, 16, 16, 1], [odinn.ACT_SOFTPLUS, odinn.ACT_SOFTPLUS, odinn.ACT_SIGMOID],
                                  [(-25.0, 0.0), (0.0, 500.0)], odinn.POST_EXPMAX, 0.0, ph.maxA)
            b.set_law(odinn.LAW_NN_Y, mlp16, np.random.default_rng(1234).uniform(-0.5, 0.5, mlp16.n_params))
            ms_nn = ev(T.TIMED_SOLVE_STEP, 3, 1)
            aux["nn_inlined_2x16_ms_per_step"] = ms_nn
            aux["nn_inlined_2x16_cellsteps_per_s"] = 5.0 * cells / (ms_nn * 1e-3)
            aux["nn_inlined_2x16_note"] = "LawY: Y = NN_theta(T, Hbar) evaluated per dual node inside the stencil (Laws.jl:258-265)"
            # the same law through its per-glacier table Y(Hbar) (odinn_schedule.law_table, the library's default inside solves and
            # gradients; the TIMED_* kernels above always evaluate the network): whole forward solves through the public API
            try:
                tsy = [2010.0, 2010.25]
                ytab = {}
                for nm_, tab_ in (("network", 0), ("table", -1)):
                    b.set_schedule(law_table=tab_)
                    b.solve(tsy, reltol=1e-8)
                    b.sync()
                    tq0 = time.perf_counter()
                    st_ = b.solve(tsy, reltol=1e-8)
                    b.sync()
                    n_att = max(1, max(s_.naccept + s_.nreject for s_ in st_))
                    ytab[nm_ + "_ms_per_step"] = (time.perf_counter() - tq0) * 1e3 / n_att
                    ytab[nm_ + "_steps"] = n_att
                info_ = b.law_table()
                ytab["cellsteps_per_s_table"] = 5.0 * cells / (ytab["table_ms_per_step"] * 1e-3)
                ytab["cellsteps_per_s_network"] = 5.0 * cells / (ytab["network_ms_per_step"] * 1e-3)
                ytab["table_usable"] = info_["usable"]
                ytab["table_max_rel_dev_from_network"] = info_["max_rel_dev"]
                ytab["note"] = ("LawY's inputs are (T_glacier, Hbar): per glacier and theta a function of Hbar alone; the stencil kernels read it from "
                                "1024 quintics per glacier built from the network (used only while the measured deviation is < 1e-12); wall-clock "
                                "per attempted step of a solve over 1/4 yr incl. the initial-step heuristic and the host loop")
                aux["y_law_table_2x16"] = ytab
                b.set_schedule()
            except Exception as e:
                aux["y_law_table_2x16"] = {"error": str(e)[:200]}
                b.set_schedule()
            # its own roofline: fp64 vector pipe, flops per cell-stage from the committed PMC pass of the per-stage kernel with this
            # network (profiles/r04/pmc_roofline.json: 64 x (ADD + MUL + 2 FMA + TRANS)_F64 per launch / cells)
            pm_nn, pm_nn_src = {}, "n/a"
            for pf in PMC_FILES:
                try:
                    pm_nn = json.load(open(pf))
                    if "rk_stage2_nnY16_8x1024" in pm_nn:
                        pm_nn_src = os.path.relpath(pf, ROOT)
                        break
                except Exception:
                    pass
            k16 = pm_nn.get("rk_stage2_nnY16_8x1024", {})
            fl = k16.get("flop_per_useful_cell_stage", 608.0 + 33.0 * 25.0 + FLOP_PER_CELL_STAGE_FALLBACK)
            ach = fl * 5.0 * cells / (ms_nn * 1e-3) / 1e12
            aux["roofline_nn_inlined"] = {
                "bound": "fp64-valu", "kernel": "k_rk_stage<LM 4> x 5 or k_rk_fused<LM 4> (2 -> 16 -> 16 -> 1 MLP inlined per dual node and stage)",
                "flop_per_cell_stage": fl, "flop_source": ("PMC: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 of k_rk_stage<2, 4> on 8 x 1024^2, "
                "%d VALU instructions per cell-stage, %.0f %% of the fp64 instructions are FMAs" % (
                    k16.get("valu_insts_per_useful_cell_stage", 0), 100.0 * k16.get("fma_share_of_f64_insts", 0.0))) if k16 else
                "model: 608 (MLP FMAs x 2) + 33 activations x 25 + stencil",
                "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "ms_per_step": ms_nn}
            # the reverse stage kernels of the per-node-network laws (default 2-3-10-3-1 net; the continuous adjoint of targets
            # :D_hybrid / :D runs five of them per reverse step): HIP events here, flops per cell-stage from the same PMC file
            radj = {}
            for nm, kind, pre, post_hi, key in (("Y", odinn.LAW_NN_Y, [(-25.0, 0.0), (0.0, 500.0)], ph.maxA, "adj_stage2_nnY_8x512"),
                                                ("U", odinn.LAW_NN_U, [(0.0, 300.0), (0.0, 0.5)], 50.0, "adj_stage2_nnU_8x512")):
                mdef = odinn.MLPSpec([2, 3, 10, 3, 1], [odinn.ACT_SOFTPLUS] * 3 + [odinn.ACT_SIGMOID], pre, odinn.POST_EXPMAX, 0.0, post_hi)
                b.set_law(kind, mdef, np.random.default_rng(1234).uniform(-0.5, 0.5, mdef.n_params))
                ms_a = ev(T.TIMED_ADJ_STAGE2, 5, 1)
                ms_f = ev(T.TIMED_RK_STAGE2, 5, 1)
                ka = pm_nn.get(key, {})
                kf = pm_nn.get(key.replace("adj_stage2", "rk_stage2"), {})
                e = {"adj_stage_ms": ms_a, "fwd_stage_ms": ms_f}
                if ka:
                    fa = ka["flop_per_useful_cell_stage"]
                    e.update({"flop_per_cell_stage": fa, "valu_insts_per_cell_stage": ka.get("valu_insts_per_useful_cell_stage"),
                              "achieved": fa * cells / (ms_a * 1e-3) / 1e12, "frac": fa * cells / (ms_a * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                              "issue_bound_ms": ka.get("valu_insts_per_useful_cell_stage", 0.0) * cells / 64.0 / 1024.0 * 5.5 / 2.4e9 * 1e3})
                if kf:
                    e["fwd_frac"] = kf["flop_per_useful_cell_stage"] * cells / (ms_f * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
                radj[nm] = e
            # the Y law through its table: the same two stage kernels in law mode LM_YTAB (schedule field law_table = 1 makes the timed
            # launches take them); structurally closed-form kernels, priced against HBM like k_rk_stage / k_adj_stage of the A-type laws
            try:
                mdef = odinn.MLPSpec([2, 3, 10, 3, 1], [odinn.ACT_SOFTPLUS] * 3 + [odinn.ACT_SIGMOID], [(-25.0, 0.0), (0.0, 500.0)],
                                     odinn.POST_EXPMAX, 0.0, ph.maxA)
                b.set_law(odinn.LAW_NN_Y, mdef, np.random.default_rng(1234).uniform(-0.5, 0.5, mdef.n_params))
                b.set_schedule(law_table=1)
                ms_at, ms_ft = ev(T.TIMED_ADJ_STAGE2, 10, 2), ev(T.TIMED_RK_STAGE2, 10, 2)
                info_ = b.law_table()
                b.set_schedule()
                radj["Y_table"] = {
                    "adj_stage_ms": ms_at, "fwd_stage_ms": ms_ft, "bound": "hbm", "table_usable": info_["usable"],
                    "adj_bytes_per_cell": 72, "adj_achieved_GBps": 72.0 * cells / (ms_at * 1e-3) / 1e9,
                    "adj_frac": 72.0 * cells / (ms_at * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "fwd_bytes_per_cell": 56, "fwd_achieved_GBps": 56.0 * cells / (ms_ft * 1e-3) / 1e9,
                    "fwd_frac": 56.0 * cells / (ms_ft * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "kernel": "k_adj_stage<2, LM_YTAB> / k_rk_stage<2, LM_YTAB>: Y(Hbar) from 1024 quintics per glacier (3 x 16 B per dual node, "
                              "L1 / L2-resident, not counted in the algorithmic bytes)"}
            except Exception as e:
                radj["Y_table"] = {"error": str(e)[:200]}
                b.set_schedule()
            # ... and the U law through the batch's bivariate table (LM_UTAB): 18 x 16 B of coefficients per dual node from a 2.4 MB table
            try:
                mdefU = odinn.MLPSpec([2, 3, 10, 3, 1], [odinn.ACT_SOFTPLUS] * 3 + [odinn.ACT_SIGMOID], [(0.0, 300.0), (0.0, 0.5)],
                                      odinn.POST_EXPMAX, 0.0, 50.0)
                b.set_law(odinn.LAW_NN_U, mdefU, np.random.default_rng(1234).uniform(-0.5, 0.5, mdefU.n_params))
                b.set_schedule(law_table=1)
                ms_au, ms_fu = ev(T.TIMED_ADJ_STAGE2, 10, 2), ev(T.TIMED_RK_STAGE2, 10, 2)
                info_ = b.law_table()
                b.set_schedule()
                radj["U_table"] = {
                    "adj_stage_ms": ms_au, "fwd_stage_ms": ms_fu, "bound": "hbm / L2 gather", "table_usable": info_["usable"],
                    "table_max_rel_dev_from_network": info_["max_rel_dev"],
                    "adj_bytes_per_cell": 72, "adj_achieved_GBps": 72.0 * cells / (ms_au * 1e-3) / 1e9,
                    "adj_frac": 72.0 * cells / (ms_au * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "fwd_bytes_per_cell": 56, "fwd_achieved_GBps": 56.0 * cells / (ms_fu * 1e-3) / 1e9,
                    "fwd_frac": 56.0 * cells / (ms_fu * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "table_patches": info_["n_intervals"],
                    "kernel": "k_adj_stage<2, LM_UTAB> / k_rk_stage<2, LM_UTAB>: U(Hbar, |grad S|) from bi-quintic patches at the coarsest of 16 x 8 "
                              "... 128 x 64 resolutions that passes the 1e-12 check (table_patches; 288 B per dual node gathered from the "
                              "table -- by the reverse kernel from the tile's patches staged in LDS -- not counted in the algorithmic bytes)"}
            except Exception as e:
                radj["U_table"] = {"error": str(e)[:200]}
                b.set_schedule()
            aux["roofline_adjoint_nn"] = {
                "bound": "fp64-valu", "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "kernel": "k_adj_stage<2, LM 3, DiscreteVJP, NK> (one RDPK3Sp35 stage of the reverse ODE, 2-3-10-3-1 network inlined per dual "
                          "node: Y law two direct evaluations, U law one perturbed five-point evaluation)",
                "laws": radj,
                "note": "issue_bound_ms = VALU instructions per cell-stage x cells / (1024 SIMDs x 64 lanes) x 5.5 cycles at 2.4 GHz: the time the "
                        "kernel's instruction stream needs at the fp64 pipe's sustained issue rate; flop counts from " + pm_nn_src}
            b.set_law(odinn.LAW_CONST_A)
        except Exception as e:
            aux["nn_inlined_error"] = str(e)[:200]

    # ---- CPU baseline (rank 0, N = 1 only): oracle C restatement on the host cores -------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import c_oracle as CO
            from oracle import sia2d_oracle as O

            H0, B, A = gl[0]
            cores = CO.lib().oc_num_threads()
            # the SAME work as the GPU's timed steps: the hoisted law as a dual-grid A field read in every stage
            # (A = minA + (maxA - minA) MLP(T), evaluated once on the host like the reference's LawA, Laws.jl:339-358)
            omlp = O.MLP(list(mlpA.widths), list(mlpA.acts), None, O.POST_AFFINE, ph.minA, ph.maxA)
            Afield = np.asfortranarray(O.mlp_eval(omlp, thetaA, temperature_field(H0, B)[None]))
            ms = CO.MultiStepper(cores, H0, B, 100.0, 100.0, O.Phys(), Afield)  # one glacier per host thread
            ms.run(1, 1e-6)
            tc0 = time.perf_counter()
            nst = 0
            while time.perf_counter() - tc0 < args.cpu_seconds:
                ms.run(2, 1e-6)
                nst += 2
            tc = time.perf_counter() - tc0
            # (i) of SURVEY 8(d): the same restatement on ONE host thread (bounded: ~3 s)
            CO.lib().oc_set_threads(1)
            st1 = CO.Stepper(H0, B, 100.0, 100.0, O.Phys(), Afield)
            st1.step(1e-6)
            t10 = time.perf_counter()
            n1 = 0
            while time.perf_counter() - t10 < 3.0:
                st1.step(1e-6)
                n1 += 1
            t1c = time.perf_counter() - t10
            CO.lib().oc_set_threads(cores)
            cpu = {
                "value": 5.0 * n * n * nst * cores / tc,
                "unit": "cell-steps/s",
                "cores": cores,
                "kind": "port",
                "value_1_thread": 5.0 * n * n * n1 / t1c,
                "sample": f"{cores} copies of ONE {n}x{n} glacier of the workload, one host thread each "
                          f"(the reference's pmap-over-glaciers pattern), {nst} RDPK3Sp35 steps each, "
                          f"oracle/sia2d_oracle.c with the gridded law hoisted into a dual-grid A field read in every stage, exactly the "
                          f"GPU's timed work (R u,B,A  W u' + the 3S*+ registers), {tc:.1f} s",
                "reference_note": "Julia reference not timed (toolchain absent): no julia binary in the image or on the GPU box; "
                                  "this is the C restatement of the same algorithm (oracle/), kind = port",
            }
        except Exception as e:  # the baseline is reported, never required
            cpu = {"value": None, "unit": "cell-steps/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}",
                   "reference_note": "Julia reference not timed (toolchain absent)"}

    # ---- roofline of the dominant kernel: fp64 VALU, flops and HBM traffic from the committed PMC passes ----
    pmc, pmc_src, pmc_scale = {}, None, 1.0
    for exact in (True, False):  # a pass on exactly this workload first; otherwise the newest one, its per-launch bytes scaled by cells
        for pf in PMC_FILES:
            try:
                pm = json.load(open(pf))
                if pmc_src is None and (pm.get("workload_cells") == cells or (not exact and pm.get("workload_cells"))):
                    pmc, pmc_src, pmc_scale = pm, os.path.relpath(pf, ROOT), cells / float(pm["workload_cells"])
            except Exception:
                pass
    kn = pmc.get("fused_step_nn_gridded", {})
    kc = pmc.get("fused_step_constA", {})
    fpcs_nn = kn.get("flop_per_executed_cell_stage", FLOP_PER_CELL_STAGE_FALLBACK)
    fpcs_c = kc.get("flop_per_executed_cell_stage", FLOP_PER_CELL_STAGE_FALLBACK)
    traffic = kn.get("hbm_bytes_per_launch")
    if traffic:
        traffic *= pmc_scale  # (per-cell figure of the committed pass x the cells of this launch when the batch differs)
    flops_useful = fpcs_nn * 5.0 * cells
    ach_tf = flops_useful / (ms_kernel_in_loop * 1e-3) / 1e12
    big = hbm.get("kernels", {}) if isinstance(hbm, dict) else {}
    r_nn = big.get("dhdt_nn_gridded")
    r_st = big.get("rk_stage2")
    if rank == 0:
        out = {
            "metric": "cell-steps/s (forward SIA2D+NN: A = NN_theta(T), fused RHS + RDPK3Sp35 stage update)",
            "value": value,
            "unit": "cell-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[4]: {G_job} synthetic {n}x{n} fp64 ice caps sharded over {world} GPU(s) ({G} on rank 0) with the configs[2] law "
                            "A = NN_theta(T) (2 hidden layers x 16 units, gridded T, hoisted once per theta as the reference's LawA); "
                            "one RDPK3Sp35 step (5 cell-steps per cell) = RK step kernel + controller, exactly odinn_solve's launch sequence for this batch "
                            "(the hoisted law itself: aux.law_field_ms once per solve, included in aux.full_solve)",
                "glaciers": G_job,
                "glaciers_per_gpu": G,
                "grid": [n, n],
                "cells": cells_job,
                "cells_per_gpu": cells,
                "hip_force_dev_kernarg": os.environ.get("HIP_FORCE_DEV_KERNARG") + " (set for this process by bench.py unless the "
                                         "environment says otherwise; the ROCm 7.2 / gfx950 default -- DESIGN.md 'Launch latency')",
                "parallelism": f"glacier-sharded x{world}, no data-path collective" + (
                    "" if world == 1 else f"; [loss, dtheta] all-reduce: {comm_note}"),
                "schedule": {"forced_fields": {k: v for k, v in sched_in_effect.items() if v != -1},
                             "odinn_env_set": sorted(k for k in os.environ if k.startswith("ODINN_")),
                             "note": "odinn_schedule in effect for the timed batch (-1 = automatic for every field not listed)"},
                "device": odinn.device_name(local),
            },
            "grad_evals_per_s": grad,
            "roofline": {
                "bound": "fp64-valu",
                "kernel": "k_rk_fused_strip<gridded A, 8 rows> (whole RDPK3Sp35 step, 5 stages temporally fused, A = NN_theta(T) field)",
                "achieved": ach_tf,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ach_tf / FP64_PEAK_TFLOPS,
                "traffic": traffic,
                "valu_busy_frac": kn.get("valu_busy_frac"),
                "valu_busy_note": "committed PMC pass: 4 x SQ_ACTIVE_INST_VALU / SIMD-cycles of the launch.  A pure v_fma_f64 stream at 4 waves "
                                  "per SIMD issues one wave-instruction per 5.3-6.0 cycles on this part (profiles/r02/valu_issue_cost_ubench.txt), "
                                  "i.e. reaches 0.67-0.75 by this measure: the kernel sits at the fp64 pipe's sustained issue rate, and only "
                                  "fewer instructions per cell-stage (now 43.8 flop in 51.6 VALU instructions, halo redundancy 1.41) move it",
                "ms_per_launch": ms_kernel_in_loop,
                "ms_per_launch_source": "HIP events on the library's stream around the fused step kernel of every one of the `steps` timed steps "
                                        "of the median repeat (odinn_bench_kernel_events): the same launches `ms_per_step` brackets, "
                                        "so kernel time <= step time; the step adds the controller launch",
                "ms_per_launch_back_to_back": ms_fused_nn,
                "flop_per_cell_stage": fpcs_nn,
                "flop_source": (pmc_src + ": 64 x (SQ_INSTS_VALU_ADD_F64 + MUL_F64 + 2 FMA_F64) per launch / executed cell-stages "
                                "(tiles x 64 x 64 x 5, halo included)") if pmc_src else
                               "fallback: profiles/r01/pmc_fused_strip_sq.md (constant-A kernel)",
                "algorithmic_flops_per_launch": flops_useful,
                "algorithmic_flops_definition": "flop_per_cell_stage x 5 stages x cells (halo recomputation is not useful work)",
                "traffic_source": (pmc_src + " (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, FETCH x 2 per the "
                                   "gfx950 note of MI355X_MICROARCH.md; committed measurement of this kernel on this workload, "
                                   "not collected in this run)") if traffic else None,
                "algorithmic_bytes_per_launch": B_PER_CELL_FUSED_NN * cells,
                "hbm_traffic_GBs": (traffic / (ms_kernel_in_loop * 1e-3) / 1e9) if traffic else None,
                "hbm_traffic_frac_of_peak": (traffic / (ms_kernel_in_loop * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "hbm_algorithmic_frac_of_peak": B_PER_CELL_FUSED_NN * cells / (ms_kernel_in_loop * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "constA_kernel": {"ms_per_launch": ms_fused_c, "flop_per_cell_stage": fpcs_c,
                                  "achieved": fpcs_c * 5.0 * cells / (ms_fused_c * 1e-3) / 1e12,
                                  "frac": fpcs_c * 5.0 * cells / (ms_fused_c * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                  "traffic": kc.get("hbm_bytes_per_launch")},
                "note": "temporal fusion took this kernel off the HBM roofline on purpose (it moves 32 B/cell per step instead of "
                        "5 x 64); its ceiling is the fp64 vector pipe.  The HBM-bound kernels of the path are roofline_hbm "
                        "(north-star stencil) and roofline_per_stage.",
            },
            "roofline_hbm": ({
                "bound": "hbm",
                "kernel": "k_dhdt, A = NN_theta(T) gridded: the fused SIA2D+NN RHS stencil (north-star target >= 40 % of the HBM roofline)",
                "achieved": r_nn["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r_nn["frac_of_hbm_peak"],
                "ms_per_launch": r_nn["ms"], "algorithmic_bytes_per_launch": B_PER_CELL_DHDT_NN * hbm["cells"],
                "working_set": hbm["working_set_note"],
                "traffic": (lambda k_: (k_.get("hbm_bytes_per_cell") or 0.0) * hbm["cells"] or None)(pmc.get("dhdt_nn_gridded_64", pmc.get("dhdt_nn_gridded_32", {}))),
                "in_infinity_cache_8_glaciers_GBs": (weak or {}).get("dhdt_nn_gridded_in_cache_GBs"),
            } if r_nn else None),
            "roofline_per_stage": ({
                "bound": "hbm",
                "kernel": "k_rk_stage<2,LM_FAST> (one RK stage of the per-stage schedule, scheme 1)",
                "achieved": r_st["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r_st["frac_of_hbm_peak"],
                "ms_per_launch": r_st["ms"], "algorithmic_bytes_per_launch": B_PER_CELL_STAGE2 * hbm["cells"],
                "working_set": hbm["working_set_note"],
                "traffic": (lambda k_: (k_.get("hbm_bytes_per_cell") or 0.0) * hbm["cells"] or None)(pmc.get("rk_stage2_64", pmc.get("rk_stage2_32", {}))),
            } if r_st else None),
            "roofline_adjoint": (lambda ka: {
                "bound": "fp64-valu",
                "kernel": "k_adj_fused_strip<constant A, dense, 7 rows> on rank 0's shard (a whole RDPK3Sp35 step of the reverse ODE of the continuous adjoint: "
                          "the reference's default gradient, gradient.jl:276-539)",
                "ms_per_launch": ms_adjf,
                "flop_per_cell_stage": ka.get("flop_per_executed_cell_stage"),
                "achieved": (ka["flop_per_executed_cell_stage"] * 5.0 * cells / (ms_adjf * 1e-3) / 1e12) if ka.get("flop_per_executed_cell_stage") else None,
                "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (ka["flop_per_executed_cell_stage"] * 5.0 * cells / (ms_adjf * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if ka.get("flop_per_executed_cell_stage") else None,
                "flop_source": "committed PMC pass: 64 x (SQ_INSTS_VALU_ADD_F64 + MUL_F64 + 2 FMA_F64) per launch / executed cell-stages (halo "
                               "redundancy %.2f not counted as useful work); %.0f VALU instructions per useful cell-stage, %.0f %% of the fp64 "
                               "instructions are FMAs" % (ka.get("halo_redundancy", 0.0), ka.get("valu_insts_per_useful_cell_stage", 0.0),
                                                          100.0 * ka.get("fma_share_of_f64_insts", 0.0)) if ka else None,
                "algorithmic_bytes_per_launch": 40.0 * cells,
                "hbm_algorithmic_GBs": 40.0 * cells / (ms_adjf * 1e-3) / 1e9,
                "hbm_algorithmic_frac_of_peak": 40.0 * cells / (ms_adjf * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": (ka.get("hbm_bytes_per_cell") or 0.0) * cells or None,
                "traffic_over_algorithmic": ka.get("ratio"),
                "valu_busy_frac": ka.get("valu_busy_frac"),
                "valu_insts_per_wave": ka.get("valu_insts_per_wave"),
                "note": "issue-bound like the forward step kernel (4230 VALU instructions per wavefront-step at the fp64 pipe's sustained issue "
                        "rate = the launch time); the 2.6 x traffic (stage-invariant H_j, dH, B re-read in stages 2-5 past a 4 MB L2) is hidden "
                        "behind it -- DESIGN.md section 5, 'what bounds the two fused kernels'",
            })(pmc.get("adj_fused_step_64", pmc.get("adj_fused_step_8", {}))),
            "cpu_baseline": cpu,
            "aux": aux,
        }
        sys.stdout.flush()
        os.dup2(_stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
        # human-readable tail (stderr): the headline next to what it must not be confused with
        rr = aux["timed_region_repeats"]
        cb = cpu if isinstance(cpu, dict) else {}
        sys.stderr.write(
            "[bench] value %.4g %s on %d GPU(s) (median of %d x %d steps; min %.4g, max %.4g), %.4f ms/step; roofline frac %.3f (%s); "
            "cpu_baseline kind=%s: %.4g %s on %s cores (%s)\n" % (
                value, out["unit"], world, rr["n"], args.steps, rr["value_min"], rr["value_max"], out["ms_per_step"],
                out["roofline"]["frac"], out["roofline"]["bound"], cb.get("kind"), cb.get("value") or float("nan"), cb.get("unit", ""),
                cb.get("cores"), "the C restatement of the oracle, NOT the Julia reference" if cb.get("kind") == "port" else "reference build"))
    b.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
