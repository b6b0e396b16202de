"""Run N launches of one timed kernel on a synthetic batch (for rocprofv3 runs).
usage: run_kernel.py <which-name> <G> <n> [iters] [law: const | nnA | nnY16 | nnY | nnU (default 2-3-10-3-1 net) | nnY_tab | nnU_tab]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier, temperature_field
T = odinn._lib
which = getattr(T, "TIMED_" + sys.argv[1].upper())
G, n = int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
law = sys.argv[5] if len(sys.argv) > 5 else "const"
gl = [make_glacier(n, k) for k in range(G)]
b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl])
ph = odinn.PhysicalParameters()
for k, (H0, B, A) in enumerate(gl):
    b.set_fields(k, H0, B)
    if law == "nnA":
        b.set_T_field(k, temperature_field(H0, B))
if law == "nnA":
    m = odinn.MLPSpec([1, 16, 16, 1], [odinn.ACT_SOFTPLUS, odinn.ACT_SOFTPLUS, odinn.ACT_SIGMOID], None, odinn.POST_AFFINE, ph.minA, ph.maxA)
    b.set_law(odinn.LAW_NN_A_GRIDDED, m, np.random.default_rng(1234).uniform(-0.5, 0.5, m.n_params))
elif law == "nnY16":
    m = odinn.MLPSpec([2, 16, 16, 1], [odinn.ACT_SOFTPLUS, odinn.ACT_SOFTPLUS, odinn.ACT_SIGMOID], [(-25.0, 0.0), (0.0, 500.0)], odinn.POST_EXPMAX, 0.0, ph.maxA)
    b.set_law(odinn.LAW_NN_Y, m, np.random.default_rng(1234).uniform(-0.5, 0.5, m.n_params))
tab = law.endswith("_tab")  # nnY_tab / nnU_tab: the law through its table (odinn_schedule.law_table = 1 makes the timed launches use it)
if tab:
    law = law[:-4]
if law in ("nnY", "nnU"):
    pre = [(-25.0, 0.0), (0.0, 500.0)] if law == "nnY" else [(0.0, 300.0), (0.0, 0.5)]
    m = odinn.MLPSpec([2, 3, 10, 3, 1], [odinn.ACT_SOFTPLUS] * 3 + [odinn.ACT_SIGMOID], pre, odinn.POST_EXPMAX, 0.0, ph.maxA if law == "nnY" else 50.0)
    b.set_law(odinn.LAW_NN_Y if law == "nnY" else odinn.LAW_NN_U, m, np.random.default_rng(1234).uniform(-0.5, 0.5, m.n_params))
if tab:
    b.set_schedule(law_table=1)
ms = b.time_kernel(which, iters=iters, warmup=2)
print(f"{sys.argv[1]} G={G} n={n} law={law}: {ms*1e3:.2f} us/launch")
b.close()
