"""Self-controlled loop vs the two-launch loop on small and medium batches (us per step of adaptive solves).
usage: sc_probe3.py  (spawns itself per setting)"""
import sys, os, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
if len(sys.argv) > 1:
    os.environ["ODINN_STEP_SC"] = sys.argv[1]
    import numpy as np
    import _odinn_import
    odinn = _odinn_import.load()
    from bench import make_glacier, alpine
    out = {}
    shapes4 = [(96, 80), (128, 112), (160, 128), (192, 160)]
    cases = [("4 alpine", [shapes4[k % 4] for k in range(4)], 50.0), ("16 alpine", [shapes4[k % 4] for k in range(16)], 50.0),
             ("64 alpine", [shapes4[k % 4] for k in range(64)], 50.0), ("128 alpine", [shapes4[k % 4] for k in range(128)], 50.0),
             ("1x256", [(256, 256)], 100.0), ("1x512", [(512, 512)], 100.0), ("4x512", [(512, 512)] * 4, 100.0),
             ("1x1024", [(1024, 1024)], 100.0), ("2x1024", [(1024, 1024)] * 2, 100.0), ("3x1024", [(1024, 1024)] * 3, 100.0),
             ("4x1024", [(1024, 1024)] * 4, 100.0)]
    for name, shapes, dx in cases:
        b = odinn.GlacierBatch(shapes, [dx] * len(shapes), A=[3e-17] * len(shapes))
        cache = {}
        for k, s in enumerate(shapes):
            if dx == 50.0:
                if s not in cache: cache[s] = alpine(*s)
                b.set_fields(k, *cache[s])
            else:
                if k % 8 not in cache: cache[k % 8] = make_glacier(s[0], k % 8)
                g = cache[k % 8]; b.set_fields(k, g[0], g[1])
        ts = [2010.0 + j / 12.0 for j in range(7)]
        st = b.solve(ts, reltol=1e-6)
        t0 = time.perf_counter()
        for _ in range(5): st = b.solve(ts, reltol=1e-6)
        dt = (time.perf_counter() - t0) / 5
        n = max(s.naccept + s.nreject for s in st)
        out[name] = (round(dt * 1e3, 3), n, round(dt * 1e6 / n, 1))
        b.close()
    print(json.dumps(out))
else:
    for sc in ("0", "1", "0", "1"):
        r = subprocess.run([sys.executable, __file__, sc], capture_output=True, text=True)
        print("ODINN_STEP_SC=" + sc, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
