"""GPU: a batch that lives through an inversion is reconfigured many times -- theta every iteration, stops, data, loss terms,
laws, kernel schedules now and then -- and the library caches what it can between calls (loss / stop tables keyed on version
counters, scratch buffers grown on demand, the hoisted law field, the interpolation scratch).  Every seed drives ONE long-lived
batch through a random sequence of such changes and, after each, compares its gradient with the gradient of a FRESH batch
configured from scratch to the same state: equal to rounding (the same kernels on the same inputs), or a cache was stale."""
import os

import numpy as np
import pytest

from conftest import rel_l2
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu
T0 = 2010.0


def _seeds():
    e = os.environ.get("ODINN_FUZZ_SEEDS")
    if e:
        a, b = e.split(":")
        return list(range(int(a), int(b)))
    return list(range(12))


class _State:
    """Everything that defines a gradient call, kept on the host; apply() configures a batch from it."""

    def __init__(self, gpu, rng):
        self.gpu, self.rng = gpu, rng
        self.G = int(rng.integers(1, 4))
        self.shapes = [(int(rng.integers(20, 70)), int(rng.integers(18, 44))) for _ in range(self.G)]
        self.dx = [float(rng.choice([40.0, 50.0, 100.0])) for _ in range(self.G)]
        self.ph = O.Phys()
        self.H0, self.B = [], []
        for (nx, ny) in self.shapes:
            x = np.linspace(-1.0, 1.0, nx)[:, None]
            y = np.linspace(-1.0, 1.0, ny)[None, :]
            H = rng.uniform(60.0, 200.0) * np.sqrt(np.maximum(0.0, 1.0 - (x / 0.7) ** 2 - (y / 0.75) ** 2))
            self.H0.append(np.asfortranarray(H))
            self.B.append(np.asfortranarray(1500.0 - 0.06 * x * nx * 25.0 + 12.0 * np.sin(3.0 * x) * np.cos(2.0 * y) + 0.0 * H))
        self.T = [float(rng.uniform(-15.0, -2.0)) for _ in range(self.G)]
        self.A = [float(rng.uniform(2e-18, 2e-17)) for _ in range(self.G)]
        self.step = 1.0 / 960.0
        self.mb = [None] * self.G
        self.loss_kind, self.comp, self.scale, self.scaling = "H", "xy", True, 1.0
        self.h_eps = self.v_eps = None
        self.dhdt_w = self.avgv_w = self.vreg_w = 0.0
        self.vjp = "discrete"
        self.sched = {}
        self.own = None
        self.new_law("scalar")
        self.new_stops(4)

    # ---- pieces that change -------------------------------------------------------------------------------------------
    def new_law(self, which):
        ph, rng, gpu = self.ph, self.rng, self.gpu
        self.law = which
        if which == "const":
            self.kind, self.gm, self.th = O.LAW_CONST_A, None, None
        elif which in ("scalar", "gridded"):
            widths, acts = [([1, 3, 10, 3, 1], [1, 1, 1, 2]), ([1, 16, 16, 1], [1, 1, 2])][int(rng.integers(0, 2))]
            self.kind = O.LAW_NN_A_SCALAR if which == "scalar" else O.LAW_NN_A_GRIDDED
            self.gm = gpu.MLPSpec(widths, acts, None, O.POST_AFFINE, ph.minA, ph.maxA)
            self.th = O.MLP(widths, acts, None, O.POST_AFFINE, ph.minA, ph.maxA).init_theta(rng)
            self.Tf = [np.asfortranarray(self.T[g] + rng.uniform(-2, 2, (s[0] - 1, s[1] - 1))) for g, s in enumerate(self.shapes)]
        elif which == "Y":
            widths, acts, pre = [2, 3, 10, 3, 1], [1, 1, 1, 2], [(-25.0, 0.0), (0.0, 500.0)]
            self.kind = O.LAW_NN_Y
            self.gm = gpu.MLPSpec(widths, acts, pre, O.POST_EXPMAX, 0.0, ph.maxA)
            self.th = O.MLP(widths, acts, pre, O.POST_EXPMAX, 0.0, ph.maxA).init_theta(rng)
            self.interp = (int(rng.integers(0, 2)), int(rng.choice([5, 20, 75])))
        else:
            widths, acts, pre = [2, 3, 10, 3, 1], [1, 1, 1, 2], [(0.0, 300.0), (0.0, 0.5)]
            self.kind = O.LAW_NN_U
            self.gm = gpu.MLPSpec(widths, acts, pre, O.POST_EXPMAX, 0.0, 50.0)
            self.th = O.MLP(widths, acts, pre, O.POST_EXPMAX, 0.0, 50.0).init_theta(rng)

    def new_stops(self, k):
        rng = self.rng
        self.ts = [T0 + j * self.step for j in range(k)]
        self.refs = [[np.asfortranarray(np.maximum(self.H0[g] * (1.0 - 0.03 * j) + (self.H0[g] > 0) * rng.normal(0, 1, self.H0[g].shape), 0.0))
                      for j in range(k)] for g in range(self.G)]
        self.tref = [list(self.ts) for _ in range(self.G)]
        self.own = None
        self.vref = None
        self.mbt = [self.ts[-1]] if any(m is not None for m in self.mb) else []
        self.dh = self.av = None
        self.dhdt_w = self.avgv_w = self.vreg_w = 0.0
        if self.loss_kind != "H":
            self.new_vrefs()

    def new_vrefs(self):
        rng = self.rng
        self.vref = []
        for g in range(self.G):
            tv = [self.ts[0], self.ts[-1]] if rng.random() < 0.5 else list(self.ts)
            maps = []
            for _ in tv:
                vx = np.asfortranarray(rng.normal(0, 20, self.shapes[g]) * (self.H0[g] > 0))
                vy = np.asfortranarray(rng.normal(0, 20, self.shapes[g]) * (self.H0[g] > 0))
                maps.append((np.asfortranarray(np.sqrt(vx ** 2 + vy ** 2)), vx, vy))
            self.vref.append((tv, maps))

    # ---- configuration of a batch ----------------------------------------------------------------------------------------
    def fresh(self):
        gpu = self.gpu
        b = gpu.GlacierBatch(self.shapes, self.dx, phys=[gpu.PhysicalParameters(**self.ph.__dict__)] * self.G, A=self.A, T=self.T)
        for g in range(self.G):
            b.set_fields(g, self.H0[g], self.B[g])
        self.push(b, everything=True)
        return b

    def push(self, b, everything=False, what=()):
        gpu = self.gpu
        W = lambda k: everything or k in what
        if W("law"):
            if self.kind == O.LAW_CONST_A:
                b.set_law(gpu.LAW_CONST_A)
            else:
                b.set_law(self.kind, self.gm, self.th)
                if self.kind == O.LAW_NN_A_GRIDDED:
                    for g in range(self.G):
                        b.set_T_field(g, self.Tf[g])
                if self.kind == O.LAW_NN_Y:
                    b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR if self.interp[0] else gpu._lib.GRAD_INTERP_NONE, self.interp[1])
        if W("refs"):
            for g in range(self.G):
                b.set_reference(g, self.tref[g], self.refs[g], 3)
        if W("vrefs") and self.vref is not None:
            for g in range(self.G):
                tv, maps = self.vref[g]
                b.set_velocity_reference(g, tv, [m[0] for m in maps], [m[1] for m in maps], [m[2] for m in maps])
        if W("loss"):
            b.set_loss({"H": gpu._lib.LOSS_H, "V": gpu._lib.LOSS_V, "HV": gpu._lib.LOSS_HV}[self.loss_kind], self.comp, self.scale, self.scaling)
            b.set_thickness_loss_function(self.h_eps)
            b.set_velocity_loss_function(self.v_eps if self.comp == "abs" else None)
        if W("mb"):
            for g in range(self.G):
                if self.mb[g] is not None:
                    m = self.mb[g]
                    b.set_mass_balance(g, m[0], m[1], m[2], m[3])
        if W("agg"):
            for g in range(self.G):
                if self.dh is not None:
                    b.set_dhdt_reference(g, *self.dh[g])
                if self.av is not None:
                    b.set_avgv_reference(g, *self.av[g])
            b.set_dhdt_loss(self.dhdt_w)
            b.set_avgv_loss(self.avgv_w, self.step, "xy")
            b.set_velocity_regularization(self.vreg_w, 2)
        if W("stops"):
            for g in range(self.G):
                b.set_glacier_stops(g, None if self.own is None else self.own[g])
        if W("vjp"):
            b.set_vjp_method(gpu._lib.VJP_CONTINUOUS if self.vjp == "continuous" else gpu._lib.VJP_DISCRETE)
        if W("sched"):
            b.set_schedule(**self.sched)

    def union(self):
        if self.own is None:
            return list(self.ts)
        return sorted(set(t for o in self.own for t in o))

    def gradient(self, b, how):
        ts = self.union()
        if how == "continuous":
            return b.loss_grad_continuous(ts, theta=self.th, mb_times=self.mbt, reltol=1e-7, adj_reltol=1e-7, adj_abstol=1e-7, n_quadrature=6)
        if how == "fixed":
            return b.loss_grad(ts, theta=self.th, mb_times=self.mbt, fixed_dt=self.step / 6.0)
        return b.loss_grad(ts, theta=self.th, mb_times=self.mbt, reltol=1e-7)

    # ---- one random change; returns what to push onto the long-lived batch -------------------------------------------------
    def mutate(self):
        rng = self.rng
        op = ["theta", "theta", "stops", "refs", "law", "loss", "mb", "agg", "own", "vjp", "sched", "log"][int(rng.integers(0, 12))]
        if op == "theta" and self.th is not None:
            self.th = self.th + 0.05 * rng.standard_normal(self.th.size)
            return op, ()
        if op == "stops":
            self.new_stops(int(rng.integers(3, 7)))
            return op, ("refs", "vrefs", "stops", "agg")
        if op == "refs":
            g = int(rng.integers(0, self.G))
            keep = sorted(rng.choice(len(self.ts), size=int(rng.integers(2, len(self.ts) + 1)), replace=False))
            if self.own is not None:
                return "theta-skip", ()
            self.tref[g] = [self.ts[j] for j in keep]
            self.refs[g] = [np.asfortranarray(self.refs[g][0] * rng.uniform(0.9, 1.0)) for _ in keep]
            return op, ("refs",)
        if op == "law":
            self.new_law(["const", "scalar", "gridded", "Y", "U"][int(rng.integers(0, 5))])
            return op, ("law",)
        if op == "loss":
            self.loss_kind = ["H", "V", "HV"][int(rng.integers(0, 3))]
            self.comp = "abs" if rng.random() < 0.4 else "xy"
            self.scale = bool(rng.random() < 0.5)
            self.scaling = float(rng.uniform(0.5, 2.0))
            if self.loss_kind != "H" and self.vref is None:
                self.new_vrefs()
            return op, ("vrefs", "loss")
        if op == "mb":
            g = int(rng.integers(0, self.G))
            S0 = self.B[g] + self.H0[g]
            ela = np.percentile(S0[self.H0[g] > 0], 60)
            k = 6e-3 * 30.0 * self.step * rng.uniform(0.5, 2.0)
            self.mb[g] = (np.asfortranarray(k * (S0 - ela)), float(k if rng.random() < 0.6 else 0.0), np.asfortranarray(S0), float(250.0 * k))
            self.mbt = [self.ts[-1]] if rng.random() < 0.5 or len(self.ts) < 4 else [self.ts[len(self.ts) // 2], self.ts[-1]]
            return op, ("mb",)
        if op == "agg" and self.own is None:
            k = len(self.ts)
            self.dh = [(self.ts[0], self.ts[int(rng.integers(1, k))], float(rng.uniform(-4, 1))) for _ in range(self.G)]
            self.av = []
            for g in range(self.G):
                i1 = int(rng.integers(0, k - 1)); i2 = int(rng.integers(i1 + 1, k))
                vx = np.asfortranarray(rng.normal(0, 15, self.shapes[g]) * (self.H0[g] > 0))
                vy = np.asfortranarray(rng.normal(0, 15, self.shapes[g]) * (self.H0[g] > 0))
                self.av.append((self.ts[i1], self.ts[i2], np.asfortranarray(np.sqrt(vx ** 2 + vy ** 2)), vx, vy))
            self.dhdt_w = float(rng.choice([0.0, 2.0])); self.avgv_w = float(rng.choice([0.0, 1.5]))
            if self.vref is None:
                self.new_vrefs()
            self.vreg_w = float(rng.choice([0.0, 30.0])) if all(len(v[0]) >= 2 for v in self.vref) else 0.0
            return op, ("vrefs", "agg")
        if op == "own" and self.G > 1 and self.dh is None and self.av is None and self.loss_kind == "H":
            self.own = []
            for g in range(self.G):
                inner = sorted(set(float(v) for v in rng.uniform(self.ts[0] + 0.2 * self.step, self.ts[-1] - 0.2 * self.step, int(rng.integers(1, 3)))))
                self.own.append([self.ts[0]] + inner + [self.ts[-1]])
                self.tref[g] = list(self.own[g])
                self.refs[g] = [np.asfortranarray(self.refs[g][0] * rng.uniform(0.9, 1.0)) for _ in self.own[g]]
            self.mbt = [self.ts[-1]] if self.mbt else []
            return op, ("refs", "stops")
        if op == "vjp":
            self.vjp = "continuous" if self.vjp == "discrete" else "discrete"
            return op, ("vjp",)
        if op == "sched":
            self.sched = [{}, dict(fused_tiles=1), dict(fused_tiles=3), dict(adj_fused=0), dict(vjph_strip=0), dict(step_sc=0), dict(adj_rows=7),
                          dict(snap_on_load=0)][int(rng.integers(0, 8))]
            return op, ("sched",)
        if op == "log":
            self.h_eps = None if self.h_eps else 0.1
            self.v_eps = None if self.v_eps else 0.1
            return op, ("loss",)
        return "none", ()


@pytest.mark.parametrize("seed", _seeds())
def test_long_lived_batch_equals_a_fresh_one_after_every_change(gpu, monkeypatch, seed):
    for key in list(os.environ):
        if key.startswith("ODINN_") and key not in ("ODINN_FUZZ_SEEDS", "ODINN_LIB"):
            monkeypatch.delenv(key, raising=False)
    rng = np.random.default_rng(424200 + seed)
    s = _State(gpu, rng)
    live = s.fresh()
    trail = []
    try:
        for it in range(14):
            op, what = s.mutate()
            trail.append(op)
            if what:
                s.push(live, what=what)
            how = ["fixed", "adaptive", "continuous"][int(rng.integers(0, 3))]
            if how != "continuous" and s.mbt and s.own is not None and any(t not in o for o in s.own for t in s.mbt):
                how = "continuous"
            try:
                Ll, gl = s.gradient(live, how)
                err_live = None
            except gpu.OdinnError as e:
                err_live = str(e)
            f = s.fresh()
            try:
                try:
                    Lf, gf = s.gradient(f, how)
                    err_fresh = None
                except gpu.OdinnError as e:
                    err_fresh = str(e)
            finally:
                f.close()
            # the same verdict (an invalid combination is refused by both, with the same message) ...
            assert (err_live is None) == (err_fresh is None), (trail, how, err_live, err_fresh)
            if err_live is not None:
                assert err_live == err_fresh, (trail, how, err_live, err_fresh)
                continue
            # ... or the same numbers: identical kernels on identical inputs
            tol = 1e-12 if how == "fixed" else 1e-9
            assert abs(Ll - Lf) <= tol * max(abs(Lf), 1e-300), (trail, how, Ll, Lf)
            if np.linalg.norm(gf) > 0:
                assert rel_l2(gl, gf) < (1e-10 if how == "fixed" else 1e-7), (trail, how, rel_l2(gl, gf))
    finally:
        live.close()
