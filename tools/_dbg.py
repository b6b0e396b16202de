import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _odinn_import
gpu = _odinn_import.load()
import test_gpu_fuzz as F
from oracle import sia2d_oracle as O
seed = int(sys.argv[1])
c = F._draw(gpu, 400000 + seed)
rng = np.random.default_rng(52000 + seed)
G, ph, kind = c["G"], c["ph"], c["kind"]
how = ["adaptive", "fixed", "euler"][int(rng.integers(0, 3))]
scheme = int(rng.choice([0, 1, 2])) if how != "euler" else gpu._lib.SCHEME_EULER_CFL
dense = int(rng.integers(0, 2))
print("kind", kind, "how", how, "scheme", scheme, "dense", dense, "sched", c["sched"], "shapes", c["shapes"], "ragged", c["ragged"], "mbt", c["mbt"], "C", ph.C)
print("mlp", getattr(c.get("gm"), "widths", None), "own", [len(o) for o in c["own"]])
for env in ({}, {"ODINN_SCHEME": "1"}, {"ODINN_SCHEME": "2"}, {"ODINN_STEP_SC": "0"}, {"ODINN_FUSED_TILES": "s"}, {"ODINN_FUSED_TILES": "t"}):
    for k in ("ODINN_SCHEME", "ODINN_STEP_SC", "ODINN_FUSED_TILES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    b = gpu.GlacierBatch(c["shapes"], c["dxs"], c["dys"], phys=[gpu.PhysicalParameters(**q.__dict__) for q in c["phs"]], A=c["As"], T=c["Ts"])
    for g in range(G):
        b.set_fields(g, c["gls"][g].H0, c["gls"][g].B)
        if c["mbs"][g] is not None:
            m = c["mbs"][g]; b.set_mass_balance(g, m.mb0, m.dmb_dS, m.S_ref, m.mb_max)
        if c["ragged"]:
            b.set_glacier_stops(g, c["own"][g])
    if kind != O.LAW_CONST_A:
        b.set_law(kind, c["gm"], c["th"])
        if kind == O.LAW_NN_A_GRIDDED:
            for g in range(G): b.set_T_field(g, c["laws"][g].T)
    if c["sched"]:
        b.set_schedule(**c["sched"])
    union = sorted(set(t for ts in c["own"] for t in ts))
    st = b.solve(union, mb_times=list(c["mbt"]), reltol=1e-8, scheme=int(os.environ.get("ODINN_SCHEME", scheme)), dense=dense)
    print(env, [(s.naccept, s.nreject) for s in st], b.law_table() if kind in (3, 4) else "")
    b.close()
for g in range(G):
    gl, law, mb = c["gls"][g], c["laws"][g], c["mbs"][g]
    cfg = O.SimConfig(tstops=c["own"][g], reltol=1e-8, mb=mb, mb_times=list(c["mbt"]) if mb is not None else ())
    snaps, so, _ = O.forward(gl, law, cfg)
    print("oracle", g, so.naccept, so.nreject, "Hmax", float(gl.H0.max()))
print("--- glacier 1 alone, and pairs")
for sel in ([1], [0, 1], [1, 2], [1, 0, 2], [2, 1, 0]):
    b = gpu.GlacierBatch([c["shapes"][g] for g in sel], [c["dxs"][g] for g in sel], [c["dys"][g] for g in sel],
                         phys=[gpu.PhysicalParameters(**c["phs"][g].__dict__) for g in sel], A=[c["As"][g] for g in sel], T=[c["Ts"][g] for g in sel])
    for k, g in enumerate(sel):
        b.set_fields(k, c["gls"][g].H0, c["gls"][g].B)
    if kind != O.LAW_CONST_A:
        b.set_law(kind, c["gm"], c["th"])
    st = b.solve(sorted(c["own"][1]), reltol=1e-8, dense=dense)
    print(sel, [(s.naccept, s.nreject, s.dt_last) for s in st])
    b.close()
print("Ts", c["Ts"], "As", c["As"], "dxs", c["dxs"], "dys", c["dys"], "n", [q.n for q in c["phs"]], "own", c["own"])
import math
for g in range(G):
    law = c["laws"][g]
    print("law", g, getattr(law, "A", None), getattr(law, "T", None) if not hasattr(getattr(law, "T", None), "shape") else "field")
