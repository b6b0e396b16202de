"""Timing of the NN-law variants (BASELINE configs[2]): hoisted gridded A = NN(T) and inlined
per-node Y = NN(T, Hbar) with a 2x16 MLP, 512^2 and 8x1024^2."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier
T = odinn._lib
cfgs = [(1, 512), (8, 1024)]
if len(sys.argv) > 1:
    cfgs = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
ph = odinn.PhysicalParameters()
rng = np.random.default_rng(1234)
for G, n in cfgs:
    gl = [make_glacier(n, k) for k in range(G)]
    for mode in ("constA", "nnA_gridded", "nnY_2x16", "nnY_default", "nnU_default"):
        b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl])
        for k, (H0, B, A) in enumerate(gl):
            b.set_fields(k, H0, B)
        if mode == "nnA_gridded":
            w, a = [1, 16, 16, 1], [1, 1, 2]
            mlp = odinn.MLPSpec(w, a, None, odinn.POST_AFFINE, ph.minA, ph.maxA)
            b.set_law(odinn.LAW_NN_A_GRIDDED, mlp, rng.uniform(-0.5, 0.5, mlp.n_params))
            for k, (H0, B, A) in enumerate(gl):
                S = B + H0
                Sd = 0.25 * (S[:-1, :-1] + S[1:, :-1] + S[:-1, 1:] + S[1:, 1:])
                b.set_T_field(k, -5.0 - 6.5e-3 * (Sd - S.mean()))
        elif mode.startswith("nnY"):
            w, a = ([2, 16, 16, 1], [1, 1, 2]) if mode == "nnY_2x16" else ([2, 3, 10, 3, 1], [1, 1, 1, 2])
            mlp = odinn.MLPSpec(w, a, [(-25.0, 0.0), (0.0, 500.0)], odinn.POST_EXPMAX, 0.0, ph.maxA)
            b.set_law(odinn.LAW_NN_Y, mlp, rng.uniform(-0.5, 0.5, mlp.n_params))
        elif mode.startswith("nnU"):
            w, a = [2, 3, 10, 3, 1], [1, 1, 1, 2]
            mlp = odinn.MLPSpec(w, a, [(0.0, 300.0), (0.0, 0.5)], odinn.POST_EXPMAX, 0.0, 50.0)
            b.set_law(odinn.LAW_NN_U, mlp, rng.uniform(-0.5, 0.5, mlp.n_params))
        cells = b.cells
        row = {"G": G, "n": n, "mode": mode}
        for w_, nm in [(T.TIMED_DHDT, "dhdt"), (T.TIMED_SOLVE_STEP, "solvestep"), (T.TIMED_VJP_H, "vjpH"), (T.TIMED_VJP_THETA, "vjpTh")]:
            try:
                ms = b.time_kernel(w_, iters=10, warmup=2)
                row[nm + "_us"] = round(ms * 1e3, 1)
            except Exception as e:
                row[nm + "_us"] = str(e)[:40]
        row["cellsteps_per_s"] = 5.0 * cells / (row["solvestep_us"] * 1e-6)
        print(json.dumps(row))
        b.close()
