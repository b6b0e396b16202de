"""Self-controlled reverse step (ODINN_ADJ_SC) vs the three-launch loop: ms per continuous-adjoint gradient over batch sizes.
usage: python tools/rsc_probe.py [alpine:G ...] [cap:n:G ...] [gridded] -- default: the set the rule in odinn_hip.hip was measured on"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from _inputs import synthetic_alpine
from bench import make_glacier, temperature_field

shapes4 = [(96, 80), (128, 112), (160, 128), (192, 160)]
ph = odinn.PhysicalParameters()
args = [a for a in sys.argv[1:] if a != "gridded"] or ["alpine:4", "alpine:16", "alpine:64", "cap:512:1", "cap:512:8", "cap:1024:2", "cap:1024:8"]
gridded = "gridded" in sys.argv[1:]


def build(spec):
    kind, *rest = spec.split(":")
    nn = odinn.NeuralNetwork(odinn.Parameters(), seed=42)
    mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA)
    if kind == "alpine":
        G = int(rest[0])
        shapes = [shapes4[k % 4] for k in range(G)]
        b = odinn.GlacierBatch(shapes, [50.0] * G, T=[-9.0 + 0.5 * (k % 7) for k in range(G)])
        cache = {}
        for k, s in enumerate(shapes):
            if s not in cache:
                cache[s] = synthetic_alpine(*s)
            b.set_fields(k, *cache[s])
        ts = [2010.0 + j / 12.0 for j in range(25)]
        fields = [cache[s] for s in shapes]
    else:
        n, G = int(rest[0]), int(rest[1])
        gl = [make_glacier(n, k) for k in range(G)]
        b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, T=[-8.0] * G)
        for k in range(G):
            b.set_fields(k, gl[k][0], gl[k][1])
        ts = [j / 12.0 for j in range(13)]
        fields = [(g[0], g[1]) for g in gl]
    if gridded:
        for k, (H0, B) in enumerate(fields):
            b.set_T_field(k, temperature_field(H0, B))
        b.set_law(odinn.LAW_NN_A_GRIDDED, mlp, nn.theta)
    else:
        b.set_law(odinn.LAW_NN_A_SCALAR, mlp, nn.theta)
    b.solve(ts, reltol=1e-8)
    for k in range(b.G):
        b.set_reference(k, ts, [b.snapshot(k, j) * (1.0 - 0.002 * j) for j in range(len(ts))], 3)
    return b, ts, odinn.NeuralNetwork(odinn.Parameters(), seed=1234).theta


for spec in args:
    row = {"case": spec + (":gridded" if gridded else "")}
    ref = None
    for sc in ("0", "1"):
        os.environ["ODINN_SCHEDULE"] = "adj_sc=" + sc
        b, ts, th0 = build(spec)
        b.loss_grad_continuous(ts, theta=th0, reltol=1e-8)
        b.sync()
        n = 3
        t0 = time.perf_counter()
        for _ in range(n):
            L, g = b.loss_grad_continuous(ts, theta=th0, reltol=1e-8)
        b.sync()
        dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            b.solve(ts, reltol=1e-8)
        b.sync()
        dtf = (time.perf_counter() - t0) / n
        rs = max(s.naccept + s.nreject for s in b.last_stats_rev)
        row["sc" + sc] = {"grad_ms": round(dt * 1e3, 3), "fwd_ms": round(dtf * 1e3, 3), "rev_steps": int(rs),
                          "us_per_rev_step": round((dt - dtf) * 1e6 / rs, 2)}
        g = np.array(g, dtype=float).ravel()
        if ref is None:
            ref = (L, g)
        else:
            row["bit_identical"] = bool(L == ref[0] and np.array_equal(g, ref[1]))
        b.close()
    print(json.dumps(row), flush=True)
