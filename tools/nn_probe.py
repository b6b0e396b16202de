"""Inlined-MLP laws: time per solve step under the per-stage schedule (ODINN_SCHEME=1) and the fused step kernel
(ODINN_SCHEME=2) for several batch shapes.  usage: nn_probe.py [law: Y16|Ydef|U]"""
import sys, os, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2:  # child: scheme, law
    os.environ["ODINN_SCHEDULE"] = "scheme=" + sys.argv[1]
    import numpy as np
    import _odinn_import
    odinn = _odinn_import.load()
    from bench import make_glacier, alpine
    T = odinn._lib
    ph = odinn.PhysicalParameters()
    lawn = sys.argv[2]
    w = {"Y16": [2, 16, 16, 1], "Ydef": [2, 3, 10, 3, 1], "U": [2, 3, 10, 3, 1]}[lawn]
    acts = [odinn.ACT_SOFTPLUS] * (len(w) - 2) + [odinn.ACT_SIGMOID]
    if lawn == "U":
        m = odinn.MLPSpec(w, acts, [(0.0, 300.0), (0.0, 0.5)], odinn.POST_EXPMAX, 0.0, 50.0); kind = odinn.LAW_NN_U
    else:
        m = odinn.MLPSpec(w, acts, [(-25.0, 0.0), (0.0, 500.0)], odinn.POST_EXPMAX, 0.0, ph.maxA); kind = odinn.LAW_NN_Y
    th = np.random.default_rng(1234).uniform(-0.5, 0.5, m.n_params)
    out = {}
    shapes4 = [(96, 80), (128, 112), (160, 128), (192, 160)]
    for name, shapes, dx in (("4 alpine", shapes4, 50.0), ("64 alpine", [shapes4[k % 4] for k in range(64)], 50.0),
                             ("1x512", [(512, 512)], 100.0), ("1x1024", [(1024, 1024)], 100.0), ("8x1024", [(1024, 1024)] * 8, 100.0)):
        b = odinn.GlacierBatch(shapes, [dx] * len(shapes))
        for k, s in enumerate(shapes):
            if dx == 50.0:
                b.set_fields(k, *alpine(*s))
            else:
                g = make_glacier(s[0], k); b.set_fields(k, g[0], g[1])
        b.set_law(kind, m, th)
        out[name] = round(b.time_kernel(T.TIMED_SOLVE_STEP, iters=5, warmup=2) * 1e3, 1)
        b.close()
    print(json.dumps(out))
else:
    law = sys.argv[1] if len(sys.argv) > 1 else "Y16"
    for scheme in ("1", "2"):
        r = subprocess.run([sys.executable, __file__, scheme, law], capture_output=True, text=True)
        print(f"law {law} scheme {scheme} (us per step):", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
