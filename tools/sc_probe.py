"""Adaptive forward solve time per step: default path vs self-controlled strip loop, over batch shapes."""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from _inputs import synthetic_alpine
from bench import make_glacier
def run(shapes, alpine):
    b = odinn.GlacierBatch(shapes, [50.0 if alpine else 100.0] * len(shapes), A=[3e-17] * len(shapes))
    for k, (nx, ny) in enumerate(shapes):
        if alpine: b.set_fields(k, *synthetic_alpine(nx, ny))
        else:
            H0, B, A = make_glacier(nx, k); b.set_fields(k, H0, B)
    ts = [2010.0 + j / 12.0 for j in range(13)]
    for _ in range(2): st = b.solve(ts, reltol=1e-6)
    t0 = time.perf_counter()
    for _ in range(5): st = b.solve(ts, reltol=1e-6)
    dt = (time.perf_counter() - t0) / 5
    n = max(s.naccept + s.nreject for s in st)
    b.close()
    return dt * 1e3, n, dt * 1e6 / n
cfgs = {"1x(96,80)": ([(96, 80)], True), "4 alpine": ([(96, 80), (128, 112), (160, 128), (192, 160)], True),
        "16 alpine": ([(96, 80), (128, 112), (160, 128), (192, 160)] * 4, True), "1x256": ([(256, 256)], False),
        "1x512": ([(512, 512)], False), "4x512": ([(512, 512)] * 4, False), "2x1024": ([(1024, 1024)] * 2, False)}
for name, (shapes, alp) in cfgs.items():
    print(name, "ms %.3f steps %d us/step %.1f" % run(shapes, alp), flush=True)
