"""Wall time per step of adaptive solves on large batches: three-launch loop (ODINN_STEP_SC=0) vs self-controlled loop (=1).
usage: sc_probe2.py  (spawns itself per setting)"""
import sys, os, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
if len(sys.argv) > 1:
    os.environ["ODINN_STEP_SC"] = sys.argv[1]
    import numpy as np
    import _odinn_import
    odinn = _odinn_import.load()
    from bench import make_glacier, temperature_field, alpine
    ph = odinn.PhysicalParameters()
    out = {}
    shapes4 = [(96, 80), (128, 112), (160, 128), (192, 160)]
    for name, shapes, dx, law in (("2x1024", [(1024, 1024)] * 2, 100.0, "const"), ("8x1024", [(1024, 1024)] * 8, 100.0, "const"),
                                  ("8x1024 nnA", [(1024, 1024)] * 8, 100.0, "nnA"), ("32x1024", [(1024, 1024)] * 32, 100.0, "const"),
                                  ("512 alpine", [shapes4[k % 4] for k in range(512)], 50.0, "const")):
        b = odinn.GlacierBatch(shapes, [dx] * len(shapes), A=[3e-17] * len(shapes))
        cache = {}
        for k, s in enumerate(shapes):
            if dx == 50.0:
                if s not in cache: cache[s] = alpine(*s)
                b.set_fields(k, *cache[s])
            else:
                if k % 8 not in cache: cache[k % 8] = make_glacier(s[0], k % 8)
                g = cache[k % 8]; b.set_fields(k, g[0], g[1])
                if law == "nnA": b.set_T_field(k, temperature_field(g[0], g[1]))
        if law == "nnA":
            m = odinn.MLPSpec([1, 16, 16, 1], [odinn.ACT_SOFTPLUS, odinn.ACT_SOFTPLUS, odinn.ACT_SIGMOID], None, odinn.POST_AFFINE, ph.minA, ph.maxA)
            b.set_law(odinn.LAW_NN_A_GRIDDED, m, np.random.default_rng(1234).uniform(-0.5, 0.5, m.n_params))
        ts = [2010.0 + j / 12.0 for j in range(7)]
        st = b.solve(ts, reltol=1e-6, dense=1)
        t0 = time.perf_counter()
        for _ in range(2): st = b.solve(ts, reltol=1e-6, dense=1)
        dt = (time.perf_counter() - t0) / 2
        n = max(s.naccept + s.nreject for s in st)
        out[name] = (round(dt * 1e3, 2), n, round(dt * 1e6 / n, 1))
        b.close()
    print(json.dumps(out))
else:
    for sc in ("0", "1"):
        r = subprocess.run([sys.executable, __file__, sc], capture_output=True, text=True)
        print(f"ODINN_STEP_SC={sc} (ms per solve, steps, us per step):", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:])
