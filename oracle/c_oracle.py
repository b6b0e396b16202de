"""ctypes loader of the oracle's C restatement (oracle/sia2d_oracle.c).  Test
infrastructure only -- see the header of sia2d_oracle.c."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle_sia2d.so")


class OcPhys(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("rho", "g", "eta0", "n", "p", "q", "C")]


_dp = C.POINTER(C.c_double)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError(f"{_PATH} missing: run `make -C oracle`")
        _lib = C.CDLL(_PATH)
        _lib.oc_num_threads.restype = C.c_int
        _lib.oc_set_threads.argtypes = [C.c_int]
        _lib.oc_sia2d_rhs.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double, C.POINTER(OcPhys),
                                      C.c_double, _dp, _dp]
        _lib.oc_rdpk3sp35_step.restype = C.c_double
        _lib.oc_rdpk3sp35_step.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double, C.POINTER(OcPhys),
                                           C.c_double, C.c_double, C.c_double, C.c_double, _dp]
        _lib.oc_multi_steps.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_dp), _dp, C.c_double, C.c_double,
                                        C.POINTER(OcPhys), C.c_double, C.c_double, C.POINTER(_dp)]
        _lib.oc_sia2d_rhs_field.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double, C.POINTER(OcPhys), _dp, _dp, _dp]
        _lib.oc_rdpk3sp35_step_field.restype = C.c_double
        _lib.oc_rdpk3sp35_step_field.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double, C.POINTER(OcPhys),
                                                 _dp, C.c_double, C.c_double, C.c_double, _dp]
        _lib.oc_multi_steps_field.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_dp), _dp, C.c_double, C.c_double,
                                              C.POINTER(OcPhys), _dp, C.c_double, C.POINTER(_dp)]
        _lib.oc_vjp_H.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, C.POINTER(OcPhys),
                                  C.c_double, _dp, _dp]
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _phys(ph):
    return OcPhys(ph.rho, ph.g, ph.eta0, ph.n, ph.p, ph.q, ph.C)


def rhs(H, B, dx, dy, ph, A):
    """A: scalar, or an (nx-1, ny-1) field on the dual grid (gridded LawA, hoisted)"""
    nx, ny = H.shape
    Hf, Bf = np.asfortranarray(H, dtype=np.float64), np.asfortranarray(B, dtype=np.float64)
    out = np.empty((nx, ny), order="F")
    work = np.empty(3 * nx * ny)
    p = _phys(ph)
    if np.ndim(A) == 2:
        Af = np.asfortranarray(A, dtype=np.float64)
        assert Af.shape == (nx - 1, ny - 1)
        lib().oc_sia2d_rhs_field(nx, ny, _p(Hf), _p(Bf), dx, dy, C.byref(p), _p(Af), _p(out), _p(work))
    else:
        lib().oc_sia2d_rhs(nx, ny, _p(Hf), _p(Bf), dx, dy, C.byref(p), A, _p(out), _p(work))
    return out


def vjp_H(lam, H, B, dx, dy, ph, A):
    nx, ny = H.shape
    Hf, Bf, Lf = (np.asfortranarray(a, dtype=np.float64) for a in (H, B, lam))
    out = np.empty((nx, ny), order="F")
    work = np.zeros(14 * nx * ny)
    p = _phys(ph)
    lib().oc_vjp_H(nx, ny, _p(Lf), _p(Hf), _p(Bf), dx, dy, C.byref(p), A, _p(out), _p(work))
    return out


class Stepper:
    """Repeated RDPK3Sp35 steps on one glacier (used by bench.py's cpu_baseline leg)."""

    def __init__(self, H0, B, dx, dy, ph, A):
        self.nx, self.ny = H0.shape
        self.u = np.asfortranarray(H0, dtype=np.float64).copy(order="F")
        self.B = np.asfortranarray(B, dtype=np.float64)
        self.dx, self.dy, self.A = dx, dy, A
        self.Af = np.asfortranarray(A, dtype=np.float64) if np.ndim(A) == 2 else None  # dual-grid A field
        self.ph = _phys(ph)
        self.work = np.empty(7 * self.nx * self.ny)

    def step(self, dt, abstol=1e-6, reltol=1e-8):
        if self.Af is not None:
            return lib().oc_rdpk3sp35_step_field(self.nx, self.ny, _p(self.u), _p(self.B), self.dx, self.dy, C.byref(self.ph),
                                                 _p(self.Af), dt, abstol, reltol, _p(self.work))
        return lib().oc_rdpk3sp35_step(self.nx, self.ny, _p(self.u), _p(self.B), self.dx, self.dy, C.byref(self.ph),
                                       self.A, dt, abstol, reltol, _p(self.work))


class MultiStepper:
    """G copies of one glacier, one host thread each (the reference's pmap-over-glaciers use of a
    multi-core host); used by bench.py's cpu_baseline leg."""

    def __init__(self, G, H0, B, dx, dy, ph, A):
        self.G = G
        self.nx, self.ny = H0.shape
        self.us = [np.asfortranarray(H0, dtype=np.float64).copy(order="F") for _ in range(G)]
        self.works = [np.empty(7 * self.nx * self.ny) for _ in range(G)]
        self.B = np.asfortranarray(B, dtype=np.float64)
        self.dx, self.dy, self.A = dx, dy, A
        self.Af = np.asfortranarray(A, dtype=np.float64) if np.ndim(A) == 2 else None
        self.ph = _phys(ph)
        self._up = (_dp * G)(*[_p(u) for u in self.us])
        self._wp = (_dp * G)(*[_p(w) for w in self.works])

    def run(self, nsteps, dt):
        if self.Af is not None:
            lib().oc_multi_steps_field(self.G, nsteps, self.nx, self.ny, self._up, _p(self.B), self.dx, self.dy,
                                       C.byref(self.ph), _p(self.Af), dt, self._wp)
            return
        lib().oc_multi_steps(self.G, nsteps, self.nx, self.ny, self._up, _p(self.B), self.dx, self.dy,
                             C.byref(self.ph), self.A, dt, self._wp)
