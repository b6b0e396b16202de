"""Generic-exponent (sliding) law and Y law: step and RHS times (integer exponents take the product-chain powers)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier
T = odinn._lib
n, G = 1024, 8
gl = [make_glacier(n, k) for k in range(G)]
out = {}
for name, kw in (("n3 C0 (fast path)", {}), ("n3 C=7e-8 p3 q0", dict(C=7e-8, p=3.0, q=0.0)), ("n3 C=7e-8 p3 q1", dict(C=7e-8, p=3.0, q=1.0)),
                 ("n3.2 C0 (pow)", dict(n=3.2))):
    ph = odinn.PhysicalParameters(**kw)
    b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, phys=[ph] * G, A=[g[2] for g in gl])
    for k, (H0, B, A) in enumerate(gl):
        b.set_fields(k, H0, B)
    out[name] = {"step_us": round(b.time_kernel(T.TIMED_SOLVE_STEP, 10, 2) * 1e3, 1), "dhdt_us": round(b.time_kernel(T.TIMED_DHDT, 10, 2) * 1e3, 1),
                 "vjpH_us": round(b.time_kernel(T.TIMED_VJP_H, 10, 2) * 1e3, 1)}
    b.close()
print(json.dumps(out))
