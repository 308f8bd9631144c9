"""ctypes wrapper of liboracle_sw2d.so (oracle_sw2d.c) -- TEST INFRASTRUCTURE ONLY.

The C file restates, pass by pass, the reference's
  blitzdg::sw2d::computeRHS        src/sw2d-simple/main.cpp:181-356
  midpoint RK2 + filter            src/sw2d-simple/main.cpp:132-151
  dt / blow-up reductions          src/sw2d-simple/main.cpp:98-109,153-167
  LSERK4 stage loop                src/advec1d/main.cpp:92-102
  advec1d::computeRHS              src/advec1d/main.cpp:126-188
Build with `make -C oracle` (done by __graft_entry__.build()).
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_double, c_int, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_sw2d.so")


class _Ctx(Structure):
    _fields_ = [("Np", c_int), ("Nfp", c_int), ("K", c_int),
                ("Dr", c_void_p), ("Ds", c_void_p), ("Lift", c_void_p), ("Filt", c_void_p),
                ("rx", c_void_p), ("sx", c_void_p), ("ry", c_void_p), ("sy", c_void_p),
                ("nx", c_void_p), ("ny", c_void_p), ("Fscale", c_void_p),
                ("vmapM", c_void_p), ("vmapP", c_void_p), ("mapW", c_void_p),
                ("nW", c_int), ("g", c_double), ("threads", c_int)]


def _lib():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} missing: run `make -C oracle`")
    lib = ctypes.CDLL(_LIB_PATH)
    lib.oracle_lserk4_a.restype = POINTER(c_double)
    lib.oracle_lserk4_b.restype = POINTER(c_double)
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(c_void_p) if a is not None else None


def lserk4_coefficients():
    lib = _lib()
    a, b = lib.oracle_lserk4_a(), lib.oracle_lserk4_b()
    return np.array([a[i] for i in range(5)]), np.array([b[i] for i in range(5)])


class Sw2dOracle:
    """Variant A (3-field) sw2d RHS + steppers on the CPU, from host tables.

    `tables` is a mapping / object with Dr, Ds, Lift, rx, sx, ry, sy, nx, ny, Fscale, vmapM,
    vmapP (reference numbering), optional Filter (`filter`), and the wall node list mapW.
    """

    def __init__(self, Dr, Ds, Lift, rx, sx, ry, sy, nx, ny, Fscale, vmapM, vmapP, mapW, g=9.81, Filter=None,
                 threads=1):
        self._lib = _lib()
        self.Dr, self.Ds, self.Lift = _f64(Dr), _f64(Ds), _f64(Lift)
        self.rx, self.sx, self.ry, self.sy = _f64(rx), _f64(sx), _f64(ry), _f64(sy)
        self.nx, self.ny, self.Fscale = _f64(nx), _f64(ny), _f64(Fscale)
        self.vmapM, self.vmapP, self.mapW = _i32(vmapM), _i32(vmapP), _i32(mapW)
        self.Filter = _f64(Filter) if Filter is not None else None
        self.Np, self.K = self.rx.shape
        self.Nfp = self.nx.shape[0] // 3
        self.g = float(g)
        self.threads = int(threads)

    def _ctx(self):
        return _Ctx(self.Np, self.Nfp, self.K, _p(self.Dr), _p(self.Ds), _p(self.Lift), _p(self.Filter),
                    _p(self.rx), _p(self.sx), _p(self.ry), _p(self.sy), _p(self.nx), _p(self.ny), _p(self.Fscale),
                    _p(self.vmapM), _p(self.vmapP), _p(self.mapW), self.mapW.size, self.g, self.threads)

    def _fields(self, *fs):
        out = [_f64(f) for f in fs]
        for f in out:
            if f.shape != (self.Np, self.K):
                raise ValueError(f"field shape {f.shape} != {(self.Np, self.K)}")
        return out

    def rhs(self, h, hu, hv, filter=False):
        h, hu, hv = self._fields(h, hu, hv)
        r = [np.empty_like(h) for _ in range(3)]
        c = self._ctx()
        fn = self._lib.oracle_sw2d_rhs_filtered if filter else self._lib.oracle_sw2d_rhs
        rc = fn(byref(c), _p(h), _p(hu), _p(hv), _p(r[0]), _p(r[1]), _p(r[2]))
        if rc:
            raise RuntimeError(f"oracle_sw2d_rhs failed ({rc})")
        return tuple(r)

    def step_rk2(self, h, hu, hv, dt, nsteps=1, filter=True):
        """Midpoint RK2 steps; returns new (h, hu, hv)."""
        h, hu, hv = [f.copy() for f in self._fields(h, hu, hv)]
        c = self._ctx()
        for _ in range(nsteps):
            rc = self._lib.oracle_sw2d_step_rk2(byref(c), _p(h), _p(hu), _p(hv), c_double(dt), int(bool(filter)))
            if rc:
                raise RuntimeError(f"oracle_sw2d_step_rk2 failed ({rc})")
        return h, hu, hv

    def step_ssprk2(self, h, hu, hv, dt, nsteps=1, filter=False, sponge=0.0):
        """Heun steps with the variant-B sponge (src/sw2d/main.cpp:211-235); returns new (h, hu, hv)."""
        h, hu, hv = [f.copy() for f in self._fields(h, hu, hv)]
        c = self._ctx()
        for _ in range(nsteps):
            rc = self._lib.oracle_sw2d_step_ssprk2(byref(c), _p(h), _p(hu), _p(hv), c_double(dt), int(bool(filter)),
                                                   c_double(sponge))
            if rc:
                raise RuntimeError(f"oracle_sw2d_step_ssprk2 failed ({rc})")
        return h, hu, hv

    def lserk4_stages(self, h, hu, hv, res, dt, first, num_stages):
        """Runs LSERK4 stages in place on copies; returns (h, hu, hv, res)."""
        h, hu, hv = [f.copy() for f in self._fields(h, hu, hv)]
        res = [f.copy() for f in self._fields(*res)]
        c = self._ctx()
        rc = self._lib.oracle_sw2d_lserk4_stages(byref(c), _p(h), _p(hu), _p(hv), _p(res[0]), _p(res[1]), _p(res[2]),
                                                 c_double(dt), int(first), int(num_stages))
        if rc:
            raise RuntimeError(f"oracle_sw2d_lserk4_stages failed ({rc})")
        return h, hu, hv, res

    def step_lserk4(self, h, hu, hv, dt, nsteps=1):
        z = [np.zeros((self.Np, self.K)) for _ in range(3)]
        h, hu, hv, _ = self.lserk4_stages(h, hu, hv, z, dt, 0, 5 * nsteps)
        return h, hu, hv

    def fsc_eta_max(self, h, hu, hv, H=None):
        h, hu, hv = self._fields(h, hu, hv)
        Hh = _f64(H) if H is not None else None
        fm, em = c_double(), c_double()
        c = self._ctx()
        rc = self._lib.oracle_sw2d_dt(byref(c), _p(h), _p(hu), _p(hv), _p(Hh), byref(fm), byref(em))
        if rc:
            raise RuntimeError(f"oracle_sw2d_dt failed ({rc})")
        return fm.value, em.value

    def dt(self, h, hu, hv, CFL, N):
        fm, _ = self.fsc_eta_max(h, hu, hv)
        return CFL / ((N + 1) * (N + 1) * 0.5 * fm)


def advec1d_rhs(Dr, Lift, rx, Fscale, nx, vmapM, vmapP, mapI, mapO, c, u):
    lib = _lib()
    Dr, Lift, rx, Fscale, nx, u = map(_f64, (Dr, Lift, rx, Fscale, nx, u))
    vmapM, vmapP = _i32(vmapM), _i32(vmapP)
    Np, K = u.shape
    out = np.empty_like(u)
    rc = lib.oracle_advec1d_rhs(Np, K, _p(Dr), _p(Lift), _p(rx), _p(Fscale), _p(nx), _p(vmapM), _p(vmapP),
                                int(mapI), int(mapO), c_double(c), _p(u), _p(out))
    if rc:
        raise RuntimeError("oracle_advec1d_rhs failed")
    return out


def advec1d_steps(Dr, Lift, rx, Fscale, nx, vmapM, vmapP, mapI, mapO, c, dt, nsteps, u):
    lib = _lib()
    Dr, Lift, rx, Fscale, nx = map(_f64, (Dr, Lift, rx, Fscale, nx))
    u = _f64(u).copy()
    vmapM, vmapP = _i32(vmapM), _i32(vmapP)
    Np, K = u.shape
    rc = lib.oracle_advec1d_steps(Np, K, _p(Dr), _p(Lift), _p(rx), _p(Fscale), _p(nx), _p(vmapM), _p(vmapP),
                                  int(mapI), int(mapO), c_double(c), c_double(dt), int(nsteps), _p(u))
    if rc:
        raise RuntimeError("oracle_advec1d_steps failed")
    return u
