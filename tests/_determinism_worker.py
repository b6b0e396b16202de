"""Worker of tests/test_gpu_determinism.py: ONE case in THIS process, one JSON line on stdout (step counts and SHA-1 digests of every
bit the case produced).  python tests/_determinism_worker.py CASE [key=value ...]   (schedule fields)"""
import sys, os, json, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _odinn_import
gpu = _odinn_import.load()
from oracle import sia2d_oracle as O


def _dig(arrs):
    h = hashlib.sha1()
    for a in arrs:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def _valleys(dxdy):
    """three ragged alpine valleys (a few strip tiles each), optionally dx != dy"""
    shapes = [(71, 26), (29, 25), (67, 38)]
    out = []
    for k, (nx, ny) in enumerate(shapes):
        H0, B = O.synthetic_valley(nx, ny, 50.0)
        H0 = np.asfortranarray(H0 * (1.0 + 1e-3 * np.random.default_rng(3 + k).random(H0.shape)))
        out.append((H0, B))
    dxs = [50.0, 100.0, 100.0]
    dys = [38.9, 98.3, 100.0] if dxdy else dxs
    return shapes, dxs, dys, out


def forward(law, sched, dxdy=True, dense=1):
    ph = O.Phys()
    shapes, dxs, dys, gl = _valleys(dxdy)
    T = [-2.0, -5.0, -8.0]
    b = gpu.GlacierBatch(shapes, dxs, dys, T=T)
    try:
        for g, (H0, B) in enumerate(gl):
            b.set_fields(g, H0, B)
        ts = [2010.0 + j / 48.0 for j in range(4)]
        own = [ts, [ts[0], ts[1] + 0.004, ts[3]], ts]
        for g in range(3):
            b.set_glacier_stops(g, own[g])
        if law == "Ag":
            om = O.default_nn(1, light=False, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
            b.set_law(O.LAW_NN_A_GRIDDED, gpu.MLPSpec(om.widths, om.acts, om.prescale, om.post_kind, om.post_lo, om.post_hi), om.init_theta(np.random.default_rng(42)))
            for g, (H0, B) in enumerate(gl):
                b.set_T_field(g, np.asfortranarray(-2.0 - 6.0 * np.random.default_rng(7 + g).random((H0.shape[0] - 1, H0.shape[1] - 1))))
        elif law == "Y":
            om = O.MLP([2, 3, 10, 3, 1], [1, 1, 1, 2], [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
            b.set_law(O.LAW_NN_Y, gpu.MLPSpec(om.widths, om.acts, om.prescale, om.post_kind, om.post_lo, om.post_hi), om.init_theta(np.random.default_rng(9)))
        if sched:
            b.set_schedule(**sched)
        union = sorted(set(t for o in own for t in o))
        st = b.solve(union, reltol=1e-8, dense=dense)
        snaps = [b.snapshot(g, j) for g in range(3) for j in range(len(own[g]))]
        return {"steps": [(s.naccept, s.nreject) for s in st], "snaps": _dig(snaps)}
    finally:
        b.close()


def reverse(law, sched):
    ph = O.Phys()
    shapes, dxs, dys, gl = _valleys(False)
    b = gpu.GlacierBatch(shapes, dxs, dys, T=[-2.0, -5.0, -8.0])
    try:
        ts = [2010.0 + j / 96.0 for j in range(4)]
        for g, (H0, B) in enumerate(gl):
            b.set_fields(g, H0, B)
            b.set_reference(g, ts, [H0 * (1.0 - 0.01 * j) for j in range(len(ts))], 3)
        if law == "A":
            om = O.default_nn(1, light=False, post_kind=O.POST_AFFINE, post_lo=ph.minA, post_hi=ph.maxA)
            th = om.init_theta(np.random.default_rng(42)); kind = O.LAW_NN_A_SCALAR
        elif law == "U":  # LawU as the reference's tests scale it: the table follows it (k_adj_fused_lds in the reverse solve)
            om = O.MLP([2, 3, 10, 3, 1], [1, 1, 1, 2], [(0.0, 300.0), (0.0, 0.5)], O.POST_EXPMAX, 0.0, 50.0)
            th = om.init_theta(np.random.default_rng(9)); kind = O.LAW_NN_U
        else:
            om = O.MLP([2, 3, 10, 3, 1], [1, 1, 1, 2], [(-25.0, 0.0), (0.0, 500.0)], O.POST_EXPMAX, 0.0, ph.maxA)
            th = om.init_theta(np.random.default_rng(9)); kind = O.LAW_NN_Y
        b.set_law(kind, gpu.MLPSpec(om.widths, om.acts, om.prescale, om.post_kind, om.post_lo, om.post_hi), th)
        if sched:
            b.set_schedule(**sched)
        L, g = b.loss_grad_continuous(ts, theta=th, reltol=1e-8, n_quadrature=8)
        rev = getattr(b, "last_stats_rev", None)
        return {"loss": float(L).hex(), "grad": _dig([g]), "lambda0": _dig([b.lambda0(k) for k in range(3)]),
                "rev_steps": [(s.naccept, s.nreject) for s in rev] if rev else None}
    finally:
        b.close()


CASES = {
    "fwdA": lambda s: forward("A", s), "fwdAg": lambda s: forward("Ag", s), "fwdY": lambda s: forward("Y", s),
    "fwdYsq": lambda s: forward("Y", s, dxdy=False), "fwdYskip": lambda s: forward("Y", s, dense=0),
    "revA": lambda s: reverse("A", s), "revY": lambda s: reverse("Y", s), "revU": lambda s: reverse("U", s),
}

if __name__ == "__main__":
    case = sys.argv[1]
    sched = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:])}
    print("RESULT " + json.dumps(CASES[case](sched)), flush=True)
