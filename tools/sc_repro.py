"""One forward solve of a test_gpu_fuzz.py::test_forward_solve draw in THIS process, under a given schedule: step counts and a digest of
the snapshots per glacier (cross-process determinism of the self-controlled step loop; fuzz seed 24379):
    python tools/sc_repro.py SEED [key=value ...] [--reps N]
run it from several fresh processes and compare the lines."""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _odinn_import
gpu = _odinn_import.load()
import test_gpu_fuzz as F
from oracle import sia2d_oracle as O


def run(seed, sched, reps=1):
    c = F._draw(gpu, 400000 + seed)
    rng = np.random.default_rng(52000 + seed)
    G, kind = c["G"], c["kind"]
    how = ["adaptive", "fixed", "euler"][int(rng.integers(0, 3))]
    scheme = int(rng.choice([0, 1, 2])) if how != "euler" else gpu._lib.SCHEME_EULER_CFL
    dense = int(rng.integers(0, 2))
    rng.choice([0.1, 0.25, 0.5])
    mbt = list(c["mbt"])
    if c["mbs"][0] is not None and rng.random() < 0.5:
        mbt = sorted(set(mbt + [c["common"][0] + 0.61 * (c["common"][-1] - c["common"][0])]))
    out = []
    for _ in range(reps):
        b = gpu.GlacierBatch(c["shapes"], c["dxs"], c["dys"], phys=[gpu.PhysicalParameters(**q.__dict__) for q in c["phs"]], A=c["As"], T=c["Ts"])
        try:
            for g in range(G):
                b.set_fields(g, c["gls"][g].H0, c["gls"][g].B)
                if c["mbs"][g] is not None:
                    m = c["mbs"][g]
                    b.set_mass_balance(g, m.mb0, m.dmb_dS, m.S_ref, m.mb_max)
                if c["ragged"]:
                    b.set_glacier_stops(g, c["own"][g])
            if kind != O.LAW_CONST_A:
                b.set_law(kind, c["gm"], c["th"])
                if kind == O.LAW_NN_A_GRIDDED:
                    for g in range(G):
                        b.set_T_field(g, c["laws"][g].T)
            sd = dict(c["sched"]); sd.update(sched)
            if sd:
                b.set_schedule(**sd)
            union = sorted(set(t for ts in c["own"] for t in ts))
            if how == "adaptive":
                st = b.solve(union, mb_times=mbt, reltol=1e-8, scheme=scheme, dense=dense)
            elif how == "fixed":
                st = b.solve(union, mb_times=mbt, fixed_dt=c["dts"], scheme=scheme, dense=dense)
            else:
                raise SystemExit("euler draw")
            dig = []
            for g in range(G):
                h = hashlib.sha1()
                for j in range(len(c["own"][g])):
                    h.update(np.ascontiguousarray(b.snapshot(g, j)).tobytes())
                dig.append(h.hexdigest()[:10])
            out.append(([(s.naccept, s.nreject) for s in st], dig))
        finally:
            b.close()
    return dict(how=how, scheme=scheme, dense=dense, kind=kind, shapes=c["shapes"], dxs=c["dxs"], dys=c["dys"], mbt=mbt,
                mb=[m is not None for m in c["mbs"]], ragged=c["ragged"], sched0=c["sched"]), out


if __name__ == "__main__":
    argv = sys.argv[1:]
    reps = 1
    if "--reps" in argv:
        i = argv.index("--reps"); reps = int(argv[i + 1]); del argv[i:i + 2]
    args = argv
    seed = int(args[0])
    sched = {}
    for a in args[1:]:
        if "=" in a:
            k, v = a.split("="); sched[k] = int(v)
    info, out = run(seed, sched, reps)
    if os.environ.get("SC_REPRO_VERBOSE"):
        print("draw", info, flush=True)
    for steps, dig in out:
        print("seed", seed, "sched", sched, "steps", steps, "digest", dig, flush=True)
