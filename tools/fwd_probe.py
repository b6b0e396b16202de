import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from _inputs import synthetic_alpine
shapes = [(96, 80), (128, 112), (160, 128), (192, 160)]
b = odinn.GlacierBatch(shapes, [50.0] * 4, A=[3e-17] * 4)
for k, (nx, ny) in enumerate(shapes):
    b.set_fields(k, *synthetic_alpine(nx, ny))
ts = [2010.0 + j / 12.0 for j in range(25)]
for _ in range(3): st = b.solve(ts, reltol=1e-8)
t0 = time.perf_counter()
for _ in range(20): st = b.solve(ts, reltol=1e-8)
print("solve ms", (time.perf_counter() - t0) / 20 * 1e3, [s.naccept + s.nreject for s in st])
ts2 = ts[:3]
t0 = time.perf_counter()
for _ in range(20): st = b.solve(ts2, reltol=1e-8)
print("solve 2 stops ms", (time.perf_counter() - t0) / 20 * 1e3, [s.naccept + s.nreject for s in st])
t0 = time.perf_counter()
for _ in range(20): st = b.solve(ts, fixed_dt=1.0 / 12.0)
print("fixed dt 24 steps ms", (time.perf_counter() - t0) / 20 * 1e3, [s.naccept + s.nreject for s in st])
