"""Launch-gap probe: per-step time of the solve-step launch sequence on small batches,
stream launches vs the same launches captured once into a hipGraph (ODINN_TIME_GRAPH=1)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from _inputs import synthetic_alpine
T = odinn._lib
shapes4 = [(96, 80), (128, 112), (160, 128), (192, 160)]
for G in [int(a) for a in (sys.argv[1:] or ["4", "64"])]:
    shapes = [shapes4[k % 4] for k in range(G)]
    b = odinn.GlacierBatch(shapes, [50.0] * G)
    for k, (nx, ny) in enumerate(shapes):
        b.set_fields(k, *synthetic_alpine(nx, ny))
    out = {"G": G}
    for name, which in (("fused_step", T.TIMED_SOLVE_STEP), ("staged_step", T.TIMED_SOLVE_STEP_STAGED), ("dhdt", T.TIMED_DHDT)):
        os.environ.pop("ODINN_TIME_GRAPH", None)
        out[name + "_us"] = b.time_kernel(which, iters=64, warmup=4) * 1e3
        os.environ["ODINN_TIME_GRAPH"] = "1"
        out[name + "_graph_us"] = b.time_kernel(which, iters=64, warmup=4) * 1e3
    os.environ.pop("ODINN_TIME_GRAPH", None)
    print(json.dumps(out))
    b.close()
