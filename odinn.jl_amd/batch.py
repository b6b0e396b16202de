"""GlacierBatch: numpy-facing wrapper of one ``odinn_batch`` (G glaciers resident on one
MI355X).  Arrays are logical ``[i, j]`` = (x, y) like the reference's Julia matrices; they
are handed to the C ABI in column-major order (i contiguous)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L


@dataclass
class PhysicalParameters:
    """Sleipnir.PhysicalParameters fields the SIA2D path reads
    (reference test/params_construction.jl:24-34; p, q are the sliding-law exponents)."""

    rho: float = 900.0
    g: float = 9.81
    eta0: float = 1.0
    n: float = 3.0
    p: float = 3.0
    q: float = 0.0
    C: float = 0.0
    minA: float = 8e-21
    maxA: float = 8e-17

    def c_struct(self):
        return L.Phys(self.rho, self.g, self.eta0, self.n, self.p, self.q, self.C, self.minA, self.maxA)


@dataclass
class MLPSpec:
    """Architecture + scaling of a Lux.Chain(Dense...) regressor
    (reference src/models/trainable_components/ML_utils.jl:23-39, target_utils.jl:58-141)."""

    widths: Sequence[int]
    acts: Sequence[int]
    prescale: Optional[Sequence[Tuple[float, float]]] = None
    post_kind: int = L.POST_NONE
    post_lo: float = 0.0
    post_hi: float = 1.0

    @property
    def n_params(self):
        return sum(self.widths[l + 1] * (self.widths[l] + 1) for l in range(len(self.acts)))

    def c_struct(self):
        d = L.MlpDesc()
        d.n_layers = len(self.acts)
        for i, w in enumerate(self.widths):
            d.widths[i] = int(w)
        for i, a in enumerate(self.acts):
            d.acts[i] = int(a)
        d.has_prescale = 1 if self.prescale is not None else 0
        if self.prescale is not None:
            for i, (lo, hi) in enumerate(self.prescale):
                d.pre_lo[i] = lo
                d.pre_hi[i] = hi
        d.post_kind = int(self.post_kind)
        d.post_lo = float(self.post_lo)
        d.post_hi = float(self.post_hi)
        return d


def _f(a, shape=None):
    a = np.asfortranarray(np.asarray(a, dtype=np.float64))
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@dataclass
class SolveStats:
    naccept: int
    nreject: int
    nrhs: int
    t_final: float
    dt_last: float


class Comm:
    """RCCL communicator behind the C ABI (odinn_comm_*): one rank per GPU, the single collective of the path is the
    sum of [loss, dtheta] over ranks (SIA2D_grad!, gradient.jl:6-31)."""

    ID_BYTES = L.COMM_ID_BYTES

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(L.COMM_ID_BYTES)
        L.check(L.lib().odinn_comm_get_unique_id(buf))
        return buf.raw

    def __init__(self, device: int, nranks: int, rank: int, uid: bytes):
        if len(uid) != L.COMM_ID_BYTES:
            raise ValueError(f"unique id must be {L.COMM_ID_BYTES} bytes")
        self._h = C.c_void_p()
        self.nranks, self.rank, self.device = nranks, rank, device
        L.check(L.lib().odinn_comm_init_rank(device, nranks, rank, C.c_char_p(uid), C.byref(self._h)))

    def rank_size(self):
        """(rank, nranks) of the communicator as the library reports them (odinn_comm_rank -> ncclCommUserRank / ncclCommCount)"""
        r, n = C.c_int(-1), C.c_int(-1)
        L.check(L.lib().odinn_comm_rank(self._h, C.byref(r), C.byref(n)))
        return int(r.value), int(n.value)

    def allreduce_sum(self, a: np.ndarray) -> np.ndarray:
        buf = np.ascontiguousarray(a, dtype=np.float64).copy()
        L.check(L.lib().odinn_comm_allreduce_sum(self._h, _p(buf), buf.size))
        return buf

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            L.lib().odinn_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GlacierBatch:
    """G glaciers on one device.  ``shapes[g] = (nx, ny)``."""

    def __init__(self, shapes, dxs, dys=None, phys=None, A=None, T=None, device=0):
        G = len(shapes)
        self.G = G
        self.shapes = [tuple(int(v) for v in s) for s in shapes]
        dys = dxs if dys is None else dys
        phys = phys if phys is not None else [PhysicalParameters() for _ in range(G)]
        if isinstance(phys, PhysicalParameters):
            phys = [phys] * G
        A = [2.21e-18] * G if A is None else list(np.broadcast_to(A, (G,)))
        T = [-5.0] * G if T is None else list(np.broadcast_to(T, (G,)))
        descs = (L.GlacierDesc * G)()
        for g in range(G):
            descs[g].nx, descs[g].ny = self.shapes[g]
            descs[g].dx = float(np.broadcast_to(dxs, (G,))[g])
            descs[g].dy = float(np.broadcast_to(dys, (G,))[g])
            descs[g].phys = phys[g].c_struct()
            descs[g].A = float(A[g])
            descs[g].T = float(T[g])
        self.phys = phys
        self._h = C.c_void_p()
        L.check(L.lib().odinn_batch_create(device, G, descs, C.byref(self._h)))
        self.P = 0
        self.law_kind = L.LAW_CONST_A
        self.tstops = None
        self._A_field = set()  # glaciers that carry a dual-grid A field (odinn_set_A_field)
        self.vjp_method = L.VJP_DISCRETE  # the library's default (odinn_set_vjp_method)

    # -- lifetime -------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            L.lib().odinn_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def cells(self):
        return int(L.lib().odinn_batch_cells(self._h))

    # -- inputs ---------------------------------------------------------------------
    def set_fields(self, g, H0, B):
        H0, B = _f(H0, self.shapes[g]), _f(B, self.shapes[g])
        L.check(L.lib().odinn_set_fields(self._h, g, _p(H0), _p(B)))

    def set_A(self, g, A):
        L.check(L.lib().odinn_set_A(self._h, g, float(A)))
        self._A_field.discard(g)  # (odinn_set_A drops the glacier's A field)

    def _dual(self, g):
        return (self.shapes[g][0] - 1, self.shapes[g][1] - 1)

    def set_A_field(self, g, A):
        A = _f(A, self._dual(g))
        L.check(L.lib().odinn_set_A_field(self._h, g, _p(A)))
        self._A_field.add(g)

    def set_T_field(self, g, T):
        T = _f(T, self._dual(g))
        L.check(L.lib().odinn_set_T_field(self._h, g, _p(T)))

    def set_law(self, kind, mlp: Optional[MLPSpec] = None, theta=None, n_H=-1.0, n_gradS=-1.0):
        if kind == L.LAW_CONST_A:
            L.check(L.lib().odinn_set_law(self._h, kind, None, None, 0, -1.0, -1.0))
            self.P = 0
        else:
            th = np.ascontiguousarray(theta, dtype=np.float64)
            d = mlp.c_struct()
            L.check(L.lib().odinn_set_law(self._h, kind, C.byref(d), _p(th), th.size, float(n_H), float(n_gradS)))
            self.P = th.size
        self.law_kind = kind

    def set_grad_interpolation(self, kind=L.GRAD_INTERP_LINEAR, n_interp_half=75):
        """`interpolation` / `n_interp_half` of SIA2D_D_hybrid_target (target_D_hybrid.jl:12-15) and SIA2D_D_target
        (target_D_pure.jl:34-39): how d law / d theta is evaluated over the dual grid in the theta-VJP of the Y law
        (knots of Hbar) and the U law (fixed node grid, bilinear).  set_law picks the reference's default."""
        L.check(L.lib().odinn_set_grad_interpolation(self._h, int(kind), int(n_interp_half)))

    def set_theta(self, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        L.check(L.lib().odinn_set_theta(self._h, _p(th), th.size))

    def set_reference(self, g, t_ref, H_ref, distance=3):
        t = np.ascontiguousarray(t_ref, dtype=np.float64)
        nx, ny = self.shapes[g]
        H = np.ascontiguousarray(np.stack([_f(h, (nx, ny)).ravel(order="F") for h in H_ref]))
        L.check(L.lib().odinn_set_reference(self._h, g, len(t), _p(t), _p(H), int(distance)))

    def set_mass_balance(self, g, mb0, dmb_dS=0.0, S_ref=None, mb_max=np.inf):
        if mb0 is None:
            L.check(L.lib().odinn_set_mass_balance(self._h, g, None, 0.0, None, np.inf))
            return
        mb0 = _f(mb0, self.shapes[g])
        sp = None
        if S_ref is not None:
            S_ref = _f(S_ref, self.shapes[g])
            sp = _p(S_ref)
        L.check(L.lib().odinn_set_mass_balance(self._h, g, _p(mb0), float(dmb_dS), sp, float(mb_max)))

    def set_velocity_reference(self, g, t_ref, Vabs, Vx, Vy):
        """glacier.velocityData: fields are nx*ny (inn1 pairing with the dual grid)."""
        t = np.ascontiguousarray(t_ref, dtype=np.float64)
        nx, ny = self.shapes[g]
        pack = lambda fs: np.ascontiguousarray(np.stack([_f(f, (nx, ny)).ravel(order="F") for f in fs]))
        a, x, y = pack(Vabs), pack(Vx), pack(Vy)
        L.check(L.lib().odinn_set_velocity_reference(self._h, g, len(t), _p(t), _p(a), _p(x), _p(y)))

    def set_loss(self, kind=L.LOSS_H, component="xy", scale_loss=True, scaling=1.0):
        """LossH | LossV(component, scale_loss) | LossHV(scaling)  (src/losses/Losses.jl)."""
        L.check(L.lib().odinn_set_loss(self._h, int(kind), 1 if component == "abs" else 0, 1 if scale_loss else 0,
                                       float(scaling)))

    def set_thickness_loss_function(self, logsum_eps=None):
        """The simple loss inside LossH / LossHV's thickness part: L2Sum (None) or LogSum(eps) (Losses.jl:34-49,207-229)."""
        if logsum_eps is None:
            L.check(L.lib().odinn_set_thickness_loss_function(self._h, L.SIMPLE_L2SUM, 0.0))
        else:
            L.check(L.lib().odinn_set_thickness_loss_function(self._h, L.SIMPLE_LOGSUM, float(logsum_eps)))

    def set_surface_velocity_factor(self, f):
        """parameters.simulation.f_surface_velocity_factor: Velocity^ = U / f for the U law (target :D)."""
        L.check(L.lib().odinn_set_surface_velocity_factor(self._h, float(f)))

    def set_velocity_loss_function(self, logsum_eps=None):
        """The simple loss inside LossV / LossHV's velocity part: L2Sum (None) or LogSum(eps) (Losses.jl:34-49,207-229;
        component :abs only -- LogSum asserts non-negative fields)."""
        if logsum_eps is None:
            L.check(L.lib().odinn_set_velocity_loss_function(self._h, L.SIMPLE_L2SUM, 0.0))
        else:
            L.check(L.lib().odinn_set_velocity_loss_function(self._h, L.SIMPLE_LOGSUM, float(logsum_eps)))

    def set_dhdt_reference(self, g, t0, t1, dhdt_ref):
        """glacier.dhdtData of LossDhdt (TimeAggregatedLosses.jl:38-113): mean elevation-change rate between t0 and t1."""
        L.check(L.lib().odinn_set_dhdt_reference(self._h, int(g), float(t0), float(t1), float(dhdt_ref)))

    def set_avgv_reference(self, g, t1, t2, vabs, vx, vy):
        """glacier.velocityData of LossAvgV (TimeAggregatedLosses.jl:115-258): ONE sample covering [t1, t2] (date1, date2),
        nx*ny arrays with V_from_H's pairing; t2 <= t1 clears it."""
        nx, ny = self.shapes[g]
        if not t2 > t1:
            L.check(L.lib().odinn_set_avgv_reference(self._h, int(g), float(t1), float(t2), None, None, None))
            return
        va, vx_, vy_ = (np.ascontiguousarray(_f(a, (nx, ny)).ravel(order="F")) for a in (vabs, vx, vy))
        L.check(L.lib().odinn_set_avgv_reference(self._h, int(g), float(t1), float(t2), _p(va), _p(vx_), _p(vy_)))

    def set_avgv_loss(self, weight=1.0, step=1.0 / 12.0, component="xy"):
        """weight of the LossAvgV term relative to the data loss (its MultiLoss lambda; 0 switches it off), the spacing of
        its time grid and the compared component (:xy | :abs)."""
        L.check(L.lib().odinn_set_avgv_loss(self._h, float(weight), float(step), 1 if component == "abs" else 0))

    def set_velocity_regularization(self, weight=1.0, distance=3):
        """VelocityRegularization (Regularization.jl:64-79,192-245): MultiLoss weight relative to the data loss (0: off) and
        the distance to the margin of its mask; evaluated at the velocity-data times of set_velocity_reference."""
        L.check(L.lib().odinn_set_velocity_regularization(self._h, float(weight), int(distance)))

    def set_dhdt_loss(self, weight=1.0):
        """weight of the LossDhdt term relative to the data loss (its MultiLoss lambda); 0 switches it off."""
        L.check(L.lib().odinn_set_dhdt_loss(self._h, float(weight)))

    def surface_V(self, g, H):
        H = _f(H, self.shapes[g])
        Vx, Vy = np.empty_like(H), np.empty_like(H)
        L.check(L.lib().odinn_surface_V(self._h, g, _p(H), _p(Vx), _p(Vy)))
        return Vx, Vy

    def surface_V_vjp_H(self, g, dVx, dVy, H):
        H, dVx, dVy = _f(H, self.shapes[g]), _f(dVx, self.shapes[g]), _f(dVy, self.shapes[g])
        out = np.empty_like(H)
        L.check(L.lib().odinn_surface_V_vjp_H(self._h, g, _p(dVx), _p(dVy), _p(H), _p(out)))
        return out

    def surface_V_vjp_theta(self, g, dVx, dVy, H):
        H, dVx, dVy = _f(H, self.shapes[g]), _f(dVx, self.shapes[g]), _f(dVy, self.shapes[g])
        P = 1 if self.law_kind == L.LAW_CONST_A else self.P
        out = np.empty(P)
        L.check(L.lib().odinn_surface_V_vjp_theta(self._h, g, _p(dVx), _p(dVy), _p(H), _p(out), P))
        return out

    # -- seams (== Huginn.SIA2D!, VJP_lambda_dSIA/dH, VJP_lambda_dSIA/dtheta) -----------
    def dhdt(self, g, H, t=0.0):
        H = _f(H, self.shapes[g])
        out = np.empty_like(H)
        L.check(L.lib().odinn_sia2d_dhdt(self._h, g, _p(H), float(t), _p(out)))
        return out

    def vjp_H(self, g, lam, H, t=0.0):
        H, lam = _f(H, self.shapes[g]), _f(lam, self.shapes[g])
        out = np.empty_like(H)
        L.check(L.lib().odinn_sia2d_vjp_H(self._h, g, _p(lam), _p(H), float(t), _p(out)))
        return out

    def vjp_theta(self, g, lam, H, t=0.0):
        H, lam = _f(H, self.shapes[g]), _f(lam, self.shapes[g])
        P = 1 if self.law_kind == L.LAW_CONST_A else self.P
        out = np.empty(P)
        L.check(L.lib().odinn_sia2d_vjp_theta(self._h, g, _p(lam), _p(H), float(t), _p(out), P))
        return out

    def mb_apply(self, g, H):
        H = _f(H, self.shapes[g])
        Hn, MB = np.empty_like(H), np.empty_like(H)
        L.check(L.lib().odinn_mb_apply(self._h, g, _p(H), _p(Hn), _p(MB)))
        return Hn, MB

    def mb_vjp_H(self, g, lam, H_pre):
        H, lam = _f(H_pre, self.shapes[g]), _f(lam, self.shapes[g])
        out = np.empty_like(H)
        L.check(L.lib().odinn_mb_vjp_H(self._h, g, _p(lam), _p(H), _p(out)))
        return out

    def eval_law(self, g, H=None):
        # one value per glacier: A = NN(T) with a scalar T, or the constant-A law without an A field (odinn_eval_law writes
        # out[0] only)
        scalar = self.law_kind == L.LAW_NN_A_SCALAR or (self.law_kind == L.LAW_CONST_A and g not in self._A_field)
        if scalar:
            out = np.empty(1)
            L.check(L.lib().odinn_eval_law(self._h, g, None, _p(out), 1))
            return float(out[0])
        nd = self._dual(g)
        out = np.empty(nd, order="F")
        Hf = _f(H if H is not None else np.zeros(self.shapes[g]), self.shapes[g])
        L.check(L.lib().odinn_eval_law(self._h, g, _p(Hf), _p(out), out.size))
        return out

    # -- device-resident time loop -----------------------------------------------------
    @staticmethod
    def _opts(reltol=1e-8, abstol=1e-6, dtmax=0.0, dt0=0.0, fixed_dt=0.0, maxiters=10 ** 6, scheme=0, dense=0, cfl=0.0):
        return L.SolverOpts(reltol, abstol, dtmax if np.isfinite(dtmax) else 0.0, dt0, fixed_dt or 0.0, maxiters,
                            int(scheme), int(dense), float(cfl))

    def solve(self, tstops, mb_times=(), **opts) -> List[SolveStats]:
        ts = np.ascontiguousarray(tstops, dtype=np.float64)
        mb = np.ascontiguousarray(mb_times, dtype=np.float64)
        o = self._opts(**opts)
        st = (L.SolveStats * self.G)()
        L.check(L.lib().odinn_solve(self._h, ts.size, _p(ts), mb.size, _p(mb) if mb.size else None, C.byref(o), st))
        self.tstops = ts
        return [SolveStats(s.naccept, s.nreject, s.nrhs, s.t_final, s.dt_last) for s in st]

    def set_schedule(self, **fields):
        """odinn_set_schedule: force kernel forms (fields of odinn_schedule, -1 / omitted = automatic); no argument: all automatic."""
        sc = L.Schedule(**fields)
        L.check(L.lib().odinn_set_schedule(self._h, C.byref(sc)))

    def get_schedule(self):
        """The schedule in effect (environment overrides applied) as a dict."""
        sc = L.Schedule()
        L.check(L.lib().odinn_get_schedule(self._h, C.byref(sc)))
        return {k: getattr(sc, k) for k in L.SCHEDULE_FIELDS}

    def law_table(self):
        """odinn_get_law_table: state of the Y law's table (schedule field law_table) as a dict."""
        use, ni, dev = C.c_int(0), C.c_int(0), np.zeros(1)
        hmax = np.zeros(self.G)
        L.check(L.lib().odinn_get_law_table(self._h, C.byref(use), C.byref(ni), _p(dev), _p(hmax)))
        return {"usable": bool(use.value), "n_intervals": ni.value, "max_rel_dev": float(dev[0]), "hmax": hmax}

    def set_glacier_stops(self, g, t=None):
        """Glacier g's own stop table (the reference builds tstops per glacier, inversion_utils.jl:487-495); None / empty clears.
        The `tstops` of solve / loss_grad* then serve the glaciers without one; first and last stop must equal theirs."""
        ts = np.ascontiguousarray([] if t is None else t, dtype=np.float64)
        L.check(L.lib().odinn_set_glacier_stops(self._h, g, len(ts), _p(ts) if len(ts) else None))

    def snapshot(self, g, istop):
        out = np.empty(self.shapes[g], order="F")
        L.check(L.lib().odinn_get_snapshot(self._h, g, int(istop), _p(out)))
        return out

    def H(self, g):
        out = np.empty(self.shapes[g], order="F")
        L.check(L.lib().odinn_get_H(self._h, g, _p(out)))
        return out

    def loss(self):
        out = np.empty(self.G)
        L.check(L.lib().odinn_loss(self._h, _p(out)))
        return out

    def loss_grad(self, tstops, theta=None, mb_times=(), **opts):
        """(loss, dtheta), both summed over the batch's glaciers
        (== SIA2D_grad! without the cross-device reduction)."""
        ts = np.ascontiguousarray(tstops, dtype=np.float64)
        mb = np.ascontiguousarray(mb_times, dtype=np.float64)
        o = self._opts(**opts)
        P = 1 if self.law_kind == L.LAW_CONST_A else self.P
        th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
        loss = C.c_double(0.0)
        dth = np.zeros(P)
        st = (L.SolveStats * self.G)()
        L.check(L.lib().odinn_loss_grad(self._h, _p(th) if th is not None else None, P, ts.size, _p(ts), mb.size,
                                        _p(mb) if mb.size else None, C.byref(o), C.byref(loss), _p(dth), st))
        self.tstops = ts
        self.last_stats = [SolveStats(s.naccept, s.nreject, s.nrhs, s.t_final, s.dt_last) for s in st]
        return float(loss.value), dth

    def loss_grad_continuous(self, tstops, theta=None, mb_times=(), adj_reltol=1e-8, adj_abstol=1e-8,
                             adj_dtmax=1.0 / 12.0, n_quadrature=200, adj_maxiters=10 ** 6, **opts):
        """(loss, dtheta) with the continuous adjoint (ContinuousAdjoint(VJP_method = DiscreteVJP()),
        gradient.jl:276-539): reverse ODE + Gauss-Legendre quadrature, all on the device."""
        ts = np.ascontiguousarray(tstops, dtype=np.float64)
        mb = np.ascontiguousarray(mb_times, dtype=np.float64)
        o = self._opts(**opts)
        ao = L.AdjointOpts(adj_reltol, adj_abstol, adj_dtmax, int(n_quadrature), 0, int(adj_maxiters))
        P = 1 if self.law_kind == L.LAW_CONST_A else self.P
        th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
        loss = C.c_double(0.0)
        dth = np.zeros(P)
        st = (L.SolveStats * self.G)()
        sr = (L.SolveStats * self.G)()
        L.check(L.lib().odinn_loss_grad_continuous(
            self._h, _p(th) if th is not None else None, P, ts.size, _p(ts), mb.size, _p(mb) if mb.size else None,
            C.byref(o), C.byref(ao), C.byref(loss), _p(dth), st, sr))
        self.tstops = ts
        self.last_stats = [SolveStats(s.naccept, s.nreject, s.nrhs, s.t_final, s.dt_last) for s in st]
        self.last_stats_rev = [SolveStats(s.naccept, s.nreject, s.nrhs, s.t_final, s.dt_last) for s in sr]
        return float(loss.value), dth

    def batch_loss_grad(self, comm: Optional["Comm"], tstops, theta=None, mb_times=(), continuous=False, adj_reltol=1e-8,
                        adj_abstol=1e-8, adj_dtmax=1.0 / 12.0, n_quadrature=200, adj_maxiters=10 ** 6, **opts):
        """odinn_batch_loss_grad == SIA2D_grad!: this rank's (loss, dtheta) summed over all ranks by ONE RCCL all-reduce
        inside the library (comm = None: single rank)."""
        ts = np.ascontiguousarray(tstops, dtype=np.float64)
        mb = np.ascontiguousarray(mb_times, dtype=np.float64)
        o = self._opts(**opts)
        ao = L.AdjointOpts(adj_reltol, adj_abstol, adj_dtmax, int(n_quadrature), 0, int(adj_maxiters))
        P = 1 if self.law_kind == L.LAW_CONST_A else self.P
        th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
        loss = C.c_double(0.0)
        dth = np.zeros(P)
        st = (L.SolveStats * self.G)()
        sr = (L.SolveStats * self.G)()
        L.check(L.lib().odinn_batch_loss_grad(
            self._h, comm._h if comm is not None else None, 1 if continuous else 0, _p(th) if th is not None else None, P,
            ts.size, _p(ts), mb.size, _p(mb) if mb.size else None, C.byref(o), C.byref(ao), C.byref(loss), _p(dth), st, sr))
        self.tstops = ts
        self.last_stats = [SolveStats(s.naccept, s.nreject, s.nrhs, s.t_final, s.dt_last) for s in st]
        if continuous:
            self.last_stats_rev = [SolveStats(s.naccept, s.nreject, s.nrhs, s.t_final, s.dt_last) for s in sr]
        return float(loss.value), dth

    def set_vjp_method(self, method=L.VJP_DISCRETE):
        """DiscreteVJP (default) or ContinuousVJP stencil for vjp_H and both adjoints (VJPTypes.jl:29-50)."""
        L.check(L.lib().odinn_set_vjp_method(self._h, int(method)))
        self.vjp_method = int(method)  # (what is in effect: api.loss_iceflow_transient restores it)

    def tikhonov(self, a, dx, dy, mask=None):
        """(loss, grad) of TikhonovRegularization(:laplacian) on one field (Regularization.jl:92-126)."""
        a = np.asfortranarray(a, dtype=np.float64)
        g = np.empty(a.shape, order="F")
        loss = C.c_double(0.0)
        m = None if mask is None else np.asfortranarray(mask, dtype=np.uint8)
        L.check(L.lib().odinn_tikhonov(self._h, a.shape[0], a.shape[1], float(dx), float(dy), _p(a),
                                       m.ctypes.data_as(C.c_void_p) if m is not None else None, C.byref(loss), _p(g)))
        return float(loss.value), g

    def grad_parts(self):
        """(loss_g, G_g = dL/dA_g) per glacier of the last loss_grad (PerGlacierModel plumbing)."""
        lg, Gg = np.empty(self.G), np.empty(self.G)
        L.check(L.lib().odinn_get_grad_parts(self._h, _p(lg), _p(Gg)))
        return lg, Gg

    def grad_field(self, g):
        """dL/dA on the dual grid of glacier g (gridded A)."""
        out = np.empty(self._dual(g), order="F")
        L.check(L.lib().odinn_get_grad_field(self._h, g, _p(out)))
        return out

    def lambda0(self, g):
        out = np.empty(self.shapes[g], order="F")
        L.check(L.lib().odinn_get_lambda0(self._h, g, _p(out)))
        return out

    # -- measurement -------------------------------------------------------------------
    def time_kernel(self, which, iters=20, warmup=3):
        """Average milliseconds per launch (HIP events on the batch's stream)."""
        ms = C.c_double(0.0)
        L.check(L.lib().odinn_time_kernel(self._h, int(which), int(warmup), int(iters), C.byref(ms)))
        return ms.value / iters

    def bench_prepare(self):
        L.check(L.lib().odinn_bench_prepare(self._h))

    def bench_enqueue(self, which, first_iter, n):
        L.check(L.lib().odinn_bench_enqueue(self._h, int(which), int(first_iter), int(n)))

    def bench_kernel_events(self, on=True):
        """HIP events around the fused step kernel of every TIMED_SOLVE_STEP that bench_enqueue launches from now on."""
        L.check(L.lib().odinn_bench_kernel_events(self._h, 1 if on else 0))

    def bench_kernel_ms(self):
        """(summed kernel milliseconds, launches) of the event pairs recorded since the last call; synchronises."""
        ms, n = C.c_double(0.0), C.c_int(0)
        L.check(L.lib().odinn_bench_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def sync(self):
        L.check(L.lib().odinn_batch_sync(self._h))
