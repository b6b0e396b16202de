"""Kernel timing sweep over grid sizes / batch sizes (HIP-event timing through the C ABI)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier
T = odinn._lib
names = {T.TIMED_DHDT: ("dhdt", 24), T.TIMED_RK_STAGE2: ("stage2", 56), T.TIMED_RK_STEP: ("rkstep", 264),
         T.TIMED_SOLVE_STEP_STAGED: ("solvestep_staged", 264), T.TIMED_FUSED_STEP: ("fusedstep", 24),
         T.TIMED_SOLVE_STEP: ("solvestep", 24), T.TIMED_VJP_H: ("vjpH", 32), T.TIMED_VJP_THETA: ("vjpTh", 24),
         T.TIMED_EULER_CFL: ("eulercfl", 24), T.TIMED_ADJ_STAGE2: ("adjstage2", 72)}
cfgs = [(1, 128), (1, 256), (1, 512), (1, 1024), (1, 2048), (8, 1024), (64, 128), (16, 512)]
if len(sys.argv) > 1:
    cfgs = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for G, n in cfgs:
    gl = [make_glacier(n, k) for k in range(G)]
    b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl])
    for k, (H0, B, A) in enumerate(gl):
        b.set_fields(k, H0, B)
    cells = b.cells
    row = {"G": G, "n": n, "cells": cells}
    for w, (nm, bpc) in names.items():
        ms = b.time_kernel(w, iters=30, warmup=5)
        row[nm + "_us"] = round(ms * 1e3, 2)
        row[nm + "_GBs"] = round(bpc * cells / (ms * 1e-3) / 1e9, 1)
        row[nm + "_ns_per_cell"] = round(ms * 1e6 / cells, 4)
    print(json.dumps(row))
    b.close()
