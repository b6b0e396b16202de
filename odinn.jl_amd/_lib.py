"""ctypes binding of libodinn_hip.so (include/odinn_hip.h).  Loading never needs a GPU;
every compute entry point fails loudly (OdinnError) when no HIP device is present --
there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os

# (HIP_FORCE_DEV_KERNARG -- kernel arguments in device memory, the runtime's launch-latency setting and its default on
#  ROCm 7.2 / gfx950 -- is process-wide: this layer does not touch the environment.  bench.py and the tests request it for
#  their own process; ODINN_REQUEST_DEV_KERNARG=1 makes the library's loader request it.)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ODINN_LIB") or os.path.join(_HERE, "csrc", "libodinn_hip.so")  # ODINN_LIB: A/B builds

MAX_LAYERS = 8
COMM_ID_BYTES = 128
MAX_WIDTH = 32

LAW_CONST_A, LAW_NN_A_SCALAR, LAW_NN_A_GRIDDED, LAW_NN_Y, LAW_NN_U = range(5)
ACT_IDENTITY, ACT_SOFTPLUS, ACT_SIGMOID, ACT_GELU, ACT_TANH, ACT_RELU = range(6)
POST_NONE, POST_AFFINE, POST_EXPMAX, POST_SCALE = range(4)
(TIMED_DHDT, TIMED_RK_STEP, TIMED_VJP_H, TIMED_VJP_THETA, TIMED_RK_STAGE2, TIMED_SOLVE_STEP, TIMED_FUSED_STEP,
 TIMED_SOLVE_STEP_STAGED, TIMED_FUSED_STEP_SKIP, TIMED_EULER_CFL, TIMED_ADJ_STAGE2, TIMED_ADJ_FUSED_STEP,
 TIMED_LAW_FIELD) = range(13)
SCHEME_AUTO, SCHEME_STAGED, SCHEME_FUSED = 0, 1, 2
LOSS_H, LOSS_V, LOSS_HV = 0, 1, 2
SIMPLE_L2SUM, SIMPLE_LOGSUM = 0, 1
VJP_DISCRETE, VJP_CONTINUOUS = 0, 1
GRAD_INTERP_NONE, GRAD_INTERP_LINEAR = 0, 1
SCHEME_AUTO, SCHEME_STAGED, SCHEME_FUSED, SCHEME_EULER_CFL = 0, 1, 2, 3


class OdinnError(RuntimeError):
    pass


class Phys(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("rho", "g", "eta0", "n", "p", "q", "C", "minA", "maxA")]


class GlacierDesc(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("dx", C.c_double), ("dy", C.c_double),
                ("phys", Phys), ("A", C.c_double), ("T", C.c_double)]


class MlpDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("widths", C.c_int32 * (MAX_LAYERS + 1)), ("acts", C.c_int32 * MAX_LAYERS),
                ("has_prescale", C.c_int32), ("pre_lo", C.c_double * 2), ("pre_hi", C.c_double * 2),
                ("post_kind", C.c_int32), ("post_lo", C.c_double), ("post_hi", C.c_double)]


class SolverOpts(C.Structure):
    _fields_ = [("reltol", C.c_double), ("abstol", C.c_double), ("dtmax", C.c_double), ("dt0", C.c_double),
                ("fixed_dt", C.c_double), ("maxiters", C.c_int64), ("scheme", C.c_int32), ("dense", C.c_int32),
                ("cfl", C.c_double)]


class SolveStats(C.Structure):
    _fields_ = [("naccept", C.c_int64), ("nreject", C.c_int64), ("nrhs", C.c_int64), ("t_final", C.c_double),
                ("dt_last", C.c_double)]


class AdjointOpts(C.Structure):
    """odinn_adjoint_opts: reverse solve of the continuous adjoint (AdjointTypes.jl:58-67)."""
    _fields_ = [("reltol", C.c_double), ("abstol", C.c_double), ("dtmax", C.c_double), ("n_quadrature", C.c_int32),
                ("reserved", C.c_int32), ("maxiters", C.c_int64)]


SCHEDULE_FIELDS = ("step_sc", "fused_tiles", "dhdt_strip", "vjph_strip", "vjpth_strip", "snap_on_load", "interp_streams",
                   "interp_batch", "lawgrad_wave", "vq_onepass", "adj_fused", "adj_skip", "adj_segs", "adj_rows", "adj_theta_fused",
                   "law_table", "interp_async", "adj_sc", "adj_ut_fused")


class Schedule(C.Structure):
    """odinn_schedule: which of the library's equivalent kernel forms run; every field -1 = automatic."""
    _fields_ = [(k, C.c_int32) for k in SCHEDULE_FIELDS] + [("reserved", C.c_int32 * 1)]

    def __init__(self, **kw):
        super().__init__()
        for k in SCHEDULE_FIELDS:
            setattr(self, k, int(kw.pop(k, -1)))
        if kw:
            raise TypeError(f"unknown schedule field(s): {sorted(kw)}")


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); the exported symbol list tests/test_abi.py checks against the header
SIGNATURES = {
    "odinn_last_error": (C.c_char_p, []),
    "odinn_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "odinn_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "odinn_batch_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(GlacierDesc), C.POINTER(_vp)]),
    "odinn_batch_destroy": (C.c_int, [_vp]),
    "odinn_batch_sync": (C.c_int, [_vp]),
    "odinn_set_schedule": (C.c_int, [_vp, C.POINTER(Schedule)]),
    "odinn_get_schedule": (C.c_int, [_vp, C.POINTER(Schedule)]),
    "odinn_get_law_table": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, _dp]),
    "odinn_set_fields": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "odinn_set_A": (C.c_int, [_vp, C.c_int, C.c_double]),
    "odinn_set_A_field": (C.c_int, [_vp, C.c_int, _dp]),
    "odinn_set_T_field": (C.c_int, [_vp, C.c_int, _dp]),
    "odinn_set_law": (C.c_int, [_vp, C.c_int, C.POINTER(MlpDesc), _dp, C.c_int, C.c_double, C.c_double]),
    "odinn_set_theta": (C.c_int, [_vp, _dp, C.c_int]),
    "odinn_set_grad_interpolation": (C.c_int, [_vp, C.c_int, C.c_int]),
    "odinn_set_reference": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_int]),
    "odinn_set_mass_balance": (C.c_int, [_vp, C.c_int, _dp, C.c_double, _dp, C.c_double]),
    "odinn_set_velocity_reference": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp]),
    "odinn_set_loss": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double]),
    "odinn_set_dhdt_reference": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, C.c_double]),
    "odinn_set_dhdt_loss": (C.c_int, [_vp, C.c_double]),
    "odinn_set_avgv_reference": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, _dp, _dp, _dp]),
    "odinn_set_avgv_loss": (C.c_int, [_vp, C.c_double, C.c_double, C.c_int]),
    "odinn_set_velocity_regularization": (C.c_int, [_vp, C.c_double, C.c_int]),
    "odinn_set_velocity_loss_function": (C.c_int, [_vp, C.c_int, C.c_double]),
    "odinn_set_surface_velocity_factor": (C.c_int, [_vp, C.c_double]),
    "odinn_set_thickness_loss_function": (C.c_int, [_vp, C.c_int, C.c_double]),
    "odinn_surface_V": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "odinn_surface_V_vjp_H": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp]),
    "odinn_surface_V_vjp_theta": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, C.c_int]),
    "odinn_sia2d_dhdt": (C.c_int, [_vp, C.c_int, _dp, C.c_double, _dp]),
    "odinn_sia2d_vjp_H": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_double, _dp]),
    "odinn_sia2d_vjp_theta": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_double, _dp, C.c_int]),
    "odinn_mb_apply": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "odinn_mb_vjp_H": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "odinn_eval_law": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int]),
    "odinn_solve": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, C.POINTER(SolverOpts), C.POINTER(SolveStats)]),
    "odinn_set_glacier_stops": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "odinn_get_snapshot": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "odinn_get_H": (C.c_int, [_vp, C.c_int, _dp]),
    "odinn_loss": (C.c_int, [_vp, _dp]),
    "odinn_loss_grad": (C.c_int, [_vp, _dp, C.c_int, C.c_int, _dp, C.c_int, _dp, C.POINTER(SolverOpts), _dp, _dp,
                                  C.POINTER(SolveStats)]),
    "odinn_set_vjp_method": (C.c_int, [_vp, C.c_int]),
    "odinn_tikhonov": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_double, _dp, C.c_void_p, _dp, _dp]),
    "odinn_loss_grad_continuous": (C.c_int, [_vp, _dp, C.c_int, C.c_int, _dp, C.c_int, _dp, C.POINTER(SolverOpts),
                                             C.POINTER(AdjointOpts), _dp, _dp, C.POINTER(SolveStats),
                                             C.POINTER(SolveStats)]),
    "odinn_get_lambda0": (C.c_int, [_vp, C.c_int, _dp]),
    "odinn_get_grad_parts": (C.c_int, [_vp, _dp, _dp]),
    "odinn_get_grad_field": (C.c_int, [_vp, C.c_int, _dp]),
    "odinn_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "odinn_comm_init_rank": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(_vp)]),
    "odinn_comm_destroy": (C.c_int, [_vp]),
    "odinn_comm_rank": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "odinn_comm_allreduce_sum": (C.c_int, [_vp, _dp, C.c_int]),
    "odinn_comm_allreduce_sum_dev": (C.c_int, [_vp, C.c_void_p, C.c_int, C.c_void_p]),
    "odinn_batch_loss_grad": (C.c_int, [_vp, _vp, C.c_int, _dp, C.c_int, C.c_int, _dp, C.c_int, _dp, C.POINTER(SolverOpts),
                                        C.POINTER(AdjointOpts), _dp, _dp, C.POINTER(SolveStats), C.POINTER(SolveStats)]),
    "odinn_time_kernel": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp]),
    "odinn_bench_prepare": (C.c_int, [_vp]),
    "odinn_bench_enqueue": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "odinn_bench_kernel_events": (C.c_int, [_vp, C.c_int]),
    "odinn_bench_kernel_ms": (C.c_int, [_vp, _dp, C.POINTER(C.c_int)]),
    "odinn_batch_cells": (C.c_int64, [_vp]),
}

_lib = None


def lib():
    """The loaded shared library (raises OdinnError with build instructions if missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OdinnError(
                f"{LIB_PATH} not found: build it with `make -C odinn.jl_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
            )
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().odinn_last_error()
        raise OdinnError(f"libodinn_hip error {rc}: {msg.decode() if msg else '?'}")


def device_count():
    n = C.c_int(0)
    lib().odinn_device_count(C.byref(n))
    return n.value


def device_name(dev=0):
    buf = C.create_string_buffer(256)
    check(lib().odinn_device_name(dev, buf, 256))
    return buf.value.decode()
