"""Run N launches of one timed kernel on a synthetic batch (for rocprofv3 runs).
usage: run_kernel.py <which-name> <G> <n> [iters]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier
T = odinn._lib
which = getattr(T, "TIMED_" + sys.argv[1].upper())
G, n = int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
gl = [make_glacier(n, k) for k in range(G)]
b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl])
for k, (H0, B, A) in enumerate(gl):
    b.set_fields(k, H0, B)
ms = b.time_kernel(which, iters=iters, warmup=2)
print(f"{sys.argv[1]} G={G} n={n}: {ms*1e3:.2f} us/launch")
b.close()
