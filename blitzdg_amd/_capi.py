"""ctypes binding of libblitzdg_hip.so (the C ABI declared in include/blitzdg_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C blitzdg_amd/csrc``.
There is no fallback: if the shared object is missing, importing this module raises.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_float, c_int, c_size_t, c_uint,
                    c_ulonglong, c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# Sharing device memory between rank processes (RCCL's intra-node transport) needs the dmabuf IPC mode on this driver stack; the
# runtime reads the variable when it initialises, which is after this module is imported. A caller's own setting wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# BDG_HIP_LIBRARY: another build of the same library (profiling builds under build/); there is still no fallback
LIB_PATH = os.environ.get("BDG_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libblitzdg_hip.so")

BDG_OK, BDG_ERR_ARGUMENT, BDG_ERR_RUNTIME, BDG_ERR_HIP, BDG_ERR_UNSTABLE = 0, 1, 2, 3, 4
BDG_F64, BDG_I32 = 0, 1
BDG_SW2D_REORDER = 1
BDG_SW2D_NODAL_GEOMETRY = 2
BDG_SW2D_KEEP_ORDER = 4

# enum values, in header order
(MESH_VERTICES, MESH_ELEMENTS, MESH_ETOE, MESH_ETOF, MESH_BCTYPE, MESH_EPART, MESH_NPART) = range(7)
(TRI_R, TRI_S, TRI_X, TRI_Y, TRI_V, TRI_VINV, TRI_DR, TRI_DS, TRI_DRW, TRI_DSW, TRI_LIFT, TRI_FILTER,
 TRI_J, TRI_RX, TRI_RY, TRI_SX, TRI_SY, TRI_NX, TRI_NY, TRI_FSCALE, TRI_FMASK, TRI_FX, TRI_FY,
 TRI_VMAPM, TRI_VMAPP, TRI_MAPP, TRI_VMAPB, TRI_MAPB, TRI_GATHER, TRI_SCATTER) = range(30)
(GAUSS_NX, GAUSS_NY, GAUSS_SJ, GAUSS_J, GAUSS_RX, GAUSS_RY, GAUSS_SX, GAUSS_SY, GAUSS_X, GAUSS_Y, GAUSS_W,
 GAUSS_INTERP, GAUSS_MAPM, GAUSS_MAPP) = range(14)
(CUB_R, CUB_S, CUB_WEIGHTS, CUB_V, CUB_RX, CUB_RY, CUB_SX, CUB_SY, CUB_J, CUB_DR, CUB_DS, CUB_MM, CUB_MMCHOL,
 CUB_X, CUB_Y, CUB_W) = range(16)
(N1D_R, N1D_X, N1D_V, N1D_VINV, N1D_DR, N1D_LIFT, N1D_J, N1D_RX, N1D_NX, N1D_FMASK, N1D_FX,
 N1D_FSCALE, N1D_ETOV, N1D_ETOE, N1D_ETOF, N1D_VMAPM, N1D_VMAPP) = range(17)


class BdgError(RuntimeError):
    """A C-ABI call returned a nonzero status."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


class NumericalInstability(BdgError):
    """BDG_ERR_UNSTABLE: the reference drivers throw 'A numerical instability has occurred!'."""


class Table(Structure):
    _fields_ = [("data", c_void_p), ("rows", c_int), ("cols", c_int), ("dtype", c_int)]


class Sw2dDesc(Structure):
    _fields_ = [("order", c_int), ("num_elements", c_int),
                ("Dr", c_void_p), ("Ds", c_void_p), ("Lift", c_void_p), ("Filter", c_void_p),
                ("rx", c_void_p), ("sx", c_void_p), ("ry", c_void_p), ("sy", c_void_p),
                ("nx", c_void_p), ("ny", c_void_p), ("Fscale", c_void_p),
                ("vmapM", c_void_p), ("vmapP", c_void_p), ("mapW", c_void_p), ("num_wall", c_int),
                ("g", c_double), ("device", c_int), ("flags", c_int),
                ("num_fields", c_int), ("sources", c_int), ("zx", c_void_p), ("zy", c_void_p),
                ("coriolis", c_void_p), ("coriolis_const", c_double), ("drag", c_double)]


class Sw2dVbDesc(Structure):
    _fields_ = [("H", c_void_p), ("Hx", c_void_p), ("Hy", c_void_p), ("mapO", c_void_p), ("num_out", c_int),
                ("drag", c_double), ("coriolis", c_double), ("tide_amplitude", c_double),
                ("tide_period", c_double), ("tide_ramp", c_double), ("sponge", c_void_p)]


class Sw2dCurvedDesc(Structure):
    _fields_ = [("order", c_int), ("num_elements", c_int), ("num_cub", c_int), ("num_gauss", c_int),
                ("V", c_void_p), ("Filter", c_void_p), ("J", c_void_p),
                ("cubV", c_void_p), ("cubDr", c_void_p), ("cubDs", c_void_p),
                ("cubW", c_void_p), ("cubrx", c_void_p), ("cubry", c_void_p), ("cubsx", c_void_p), ("cubsy", c_void_p),
                ("gaussInterp", c_void_p), ("gaussW", c_void_p), ("gaussnx", c_void_p), ("gaussny", c_void_p),
                ("gmapM", c_void_p), ("gmapP", c_void_p), ("gmapW", c_void_p), ("num_wall", c_int),
                ("curvedEls", c_void_p), ("num_curved", c_int), ("MMChol", c_void_p),
                ("zx", c_void_p), ("zy", c_void_p), ("coriolis", c_void_p), ("coriolis_const", c_double),
                ("drag", c_void_p), ("drag_const", c_double), ("g", c_double), ("device", c_int), ("flags", c_int)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C blitzdg_amd/csrc` (there is no CPU fallback for the HIP path)")
    return ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)


lib = _load()

_P = c_void_p
_SIGNATURES = {
    "bdg_last_error": (c_char_p, []),
    "bdg_version": (c_int, []),
    "bdg_vandermonde1d": (c_int, [_P, c_int, c_int, _P, _P]),
    "bdg_mesh_create": (c_int, [POINTER(_P)]),
    "bdg_mesh_destroy": (None, [_P]),
    "bdg_mesh_read": (c_int, [_P, c_char_p]),
    "bdg_mesh_write": (c_int, [_P, c_char_p]),
    "bdg_mesh_write_cache": (c_int, [_P, c_char_p]),
    "bdg_mesh_read_cache": (c_int, [_P, c_char_p]),
    "bdg_mesh_build": (c_int, [_P, _P, c_int, _P, c_int, c_int]),
    "bdg_mesh_build_box": (c_int, [_P, c_int, c_int, c_double, c_double, c_double, c_double, c_ulonglong]),
    "bdg_mesh_set_bctype": (c_int, [_P, _P, c_int]),
    "bdg_mesh_partition": (c_int, [_P, c_int]),
    "bdg_mesh_num_elements": (c_int, [_P]),
    "bdg_mesh_num_verts": (c_int, [_P]),
    "bdg_mesh_table": (c_int, [_P, c_int, POINTER(Table)]),
    "bdg_trinodes_create": (c_int, [c_int, _P, POINTER(_P)]),
    "bdg_trinodes_destroy": (None, [_P]),
    "bdg_trinodes_build_filter": (c_int, [_P, c_double, c_int]),
    "bdg_trinodes_build_bchash": (c_int, [_P, _P, c_int]),
    "bdg_trinodes_set_coordinates": (c_int, [_P, _P, _P]),
    "bdg_trinodes_dims": (c_int, [_P, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "bdg_trinodes_table": (c_int, [_P, c_int, POINTER(Table)]),
    "bdg_trinodes_bcmap_num_tags": (c_int, [_P]),
    "bdg_trinodes_bcmap_tags": (c_int, [_P, POINTER(c_int), c_int]),
    "bdg_trinodes_bcmap_nodes": (c_int, [_P, c_int, POINTER(POINTER(c_int)), POINTER(c_int)]),
    "bdg_trinodes_split_count": (c_int, [_P]),
    "bdg_trinodes_split_operators": (c_int, [_P, _P, _P]),
    "bdg_trinodes_split_elements": (c_int, [_P, _P, _P, _P, _P]),
    "bdg_trinodes_write_vtu": (c_int, [_P, c_char_p, _P, c_char_p]),
    "bdg_trinodes_build_gauss_face_nodes": (c_int, [_P, c_int, POINTER(_P)]),
    "bdg_gaussctx_destroy": (None, [_P]),
    "bdg_gaussctx_ngauss": (c_int, [_P]),
    "bdg_gaussctx_table": (c_int, [_P, c_int, POINTER(Table)]),
    "bdg_gaussctx_bcmap_num_tags": (c_int, [_P]),
    "bdg_gaussctx_bcmap_tags": (c_int, [_P, POINTER(c_int), c_int]),
    "bdg_gaussctx_bcmap_nodes": (c_int, [_P, c_int, POINTER(POINTER(c_int)), POINTER(c_int)]),
    "bdg_trinodes_build_cubature_volume_mesh": (c_int, [_P, c_int, POINTER(_P)]),
    "bdg_cubature_rule_num_points": (c_int, [c_int]),
    "bdg_cubature_rule": (c_int, [c_int, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "bdg_cubctx_destroy": (None, [_P]),
    "bdg_cubctx_num_points": (c_int, [_P]),
    "bdg_cubctx_order": (c_int, [_P]),
    "bdg_cubctx_table": (c_int, [_P, c_int, POINTER(Table)]),
    "bdg_nodes1d_create": (c_int, [c_int, c_int, c_double, c_double, POINTER(_P)]),
    "bdg_nodes1d_destroy": (None, [_P]),
    "bdg_nodes1d_build_nodes": (c_int, [_P]),
    "bdg_nodes1d_compute_jacobian": (c_int, [_P]),
    "bdg_nodes1d_map_i": (c_int, [_P]),
    "bdg_nodes1d_map_o": (c_int, [_P]),
    "bdg_nodes1d_table": (c_int, [_P, c_int, POINTER(Table)]),
    "bdg_lserk4_num_stages": (c_int, []),
    "bdg_lserk4_a": (POINTER(c_double), []),
    "bdg_lserk4_b": (POINTER(c_double), []),
    "bdg_nodes1d_advec_rhs": (c_int, [_P, _P, c_double, _P]),
    "bdg_nodes1d_burgers_rhs": (c_int, [_P, _P, c_double, c_double, c_double, c_double, _P]),
    "bdg_burgers1d_run": (c_int, [c_int, c_int, c_double, c_double, c_double, c_double, c_double, c_double, c_double,
                                  POINTER(c_double), POINTER(c_int)]),
    "bdg_advec1d_run": (c_int, [c_int, c_int, c_double, c_double, c_double, c_double, c_double,
                                POINTER(c_double), POINTER(c_int)]),
    "bdg_device_count": (c_int, []),
    "bdg_sw2d_create": (c_int, [POINTER(Sw2dDesc), POINTER(_P)]),
    "bdg_sw2d_create_from_nodes": (c_int, [_P, c_double, c_int, c_int, POINTER(_P)]),
    "bdg_sw2d_destroy": (None, [_P]),
    "bdg_sw2d_set_state": (c_int, [_P, _P, _P, _P]),
    "bdg_sw2d_get_state": (c_int, [_P, _P, _P, _P]),
    "bdg_sw2d_set_bathymetry": (c_int, [_P, _P]),
    "bdg_sw2d_output_fields": (c_int, [_P, _P, _P, _P, _P]),
    "bdg_sw2d_output_tracer": (c_int, [_P, _P, _P]),
    "bdg_write_vtu_triangles": (c_int, [c_char_p, _P, _P, _P, c_int, c_char_p]),
    "bdg_sw2d_rhs": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int]),
    "bdg_sw2d_set_state4": (c_int, [_P, _P, _P, _P, _P]),
    "bdg_sw2d_get_state4": (c_int, [_P, _P, _P, _P, _P]),
    "bdg_sw2d_rhs4": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int]),
    "bdg_sw2d_num_fields": (c_int, [_P]),
    "bdg_sw2d_enable_variant_b": (c_int, [_P, POINTER(Sw2dVbDesc)]),
    "bdg_sw2d_set_time": (c_int, [_P, c_double]),
    "bdg_sw2d_get_time": (c_int, [_P, POINTER(c_double)]),
    "bdg_sw2d_global_speed": (c_int, [_P, POINTER(c_double)]),
    "bdg_trinodes_bed_slopes": (c_int, [_P, _P, _P, _P]),
    "bdg_trinodes_sponge_coeff": (c_int, [_P, _P, c_int, c_double, c_double, _P]),
    "bdg_sw2d_step_lserk4": (c_int, [_P, c_double, c_int]),
    "bdg_sw2d_lserk4_stages": (c_int, [_P, c_double, c_int]),
    "bdg_sw2d_step_rk2": (c_int, [_P, c_double, c_int, c_int]),
    "bdg_sw2d_step_ssprk2": (c_int, [_P, c_double, c_int, c_int, c_double]),
    "bdg_sw2d_compute_dt": (c_int, [_P, c_double, POINTER(c_double), POINTER(c_double)]),
    "bdg_sw2d_run_adaptive": (c_int, [_P, c_double, c_double, c_int, c_int, POINTER(c_double),
                                      POINTER(c_double), POINTER(c_int)]),
    "bdg_sw2d_set_partition": (c_int, [_P, c_int, c_int, _P, c_int]),
    "bdg_sw2d_halo_doubles_per_element": (c_int, [_P]),
    "bdg_sw2d_halo_pack": (c_int, [_P, _P]),
    "bdg_sw2d_halo_unpack": (c_int, [_P, _P]),
    "bdg_sw2d_lserk4_stage_part": (c_int, [_P, c_double, c_int]),
    "bdg_comm_unique_id": (c_int, [_P, c_int]),
    "bdg_sw2d_comm_init": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int]),
    "bdg_sw2d_lserk4_stages_exchanged": (c_int, [_P, c_double, c_int]),
    "bdg_sw2d_step_rk2_exchanged": (c_int, [_P, c_double, c_int, c_int]),
    "bdg_sw2d_step_ssprk2_exchanged": (c_int, [_P, c_double, c_int, c_int, c_double]),
    "bdg_sw2d_local_peers": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int]),
    "bdg_sw2d_group_lserk4_stages": (c_int, [POINTER(_P), c_int, c_double, c_int]),
    "bdg_sw2d_compute_dt_global": (c_int, [_P, c_double, POINTER(c_double), POINTER(c_double)]),
    "bdg_sw2d_allreduce_max": (c_int, [_P, c_double, POINTER(c_double)]),
    "bdg_sw2d_allreduce_sum": (c_int, [_P, c_double, POINTER(c_double)]),
    "bdg_sw2d_barrier": (c_int, [_P]),
    "bdg_sw2d_rhs_resident": (c_int, [_P, _P, _P, _P]),
    "bdg_sw2d_synchronize": (c_int, [_P]),
    "bdg_sw2d_time_lserk4_stages": (c_int, [_P, c_double, c_int, POINTER(c_float)]),
    "bdg_probe_stream_triad": (c_int, [c_int, c_size_t, c_int, POINTER(c_double)]),
    "bdg_sw2d_probe_stage_traffic": (c_int, [_P, c_int, POINTER(c_float)]),
    "bdg_sw2d_uses_affine_geometry": (c_int, [_P]),
    "bdg_sw2d_is_renumbered": (c_int, [_P]),
    "bdg_sw2d_device_bytes": (c_size_t, [_P]),
    "bdg_sw2d_stream": (c_void_p, [_P]),
    "bdg_sw2d_curved_create": (c_int, [POINTER(Sw2dCurvedDesc), POINTER(_P)]),
    "bdg_sw2d_curved_destroy": (None, [_P]),
    "bdg_sw2d_curved_rhs": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int]),
    "bdg_sw2d_curved_set_state": (c_int, [_P, _P, _P, _P, _P]),
    "bdg_sw2d_curved_get_state": (c_int, [_P, _P, _P, _P, _P]),
    "bdg_sw2d_curved_step_rk2": (c_int, [_P, c_double, c_int, c_int]),
    "bdg_sw2d_curved_lserk4_stages": (c_int, [_P, c_double, c_int]),
    "bdg_sw2d_curved_time_rk2": (c_int, [_P, c_double, c_int, c_int, POINTER(c_float)]),
    "bdg_sw2d_curved_synchronize": (c_int, [_P]),
    "bdg_sw2d_curved_device_bytes": (c_size_t, [_P]),
    "bdg_sw2d_curved_bytes_per_element": (c_double, [_P]),
    "bdg_sw2d_curved_form": (c_int, [_P]),
    "bdg_sw2d_curved_get_elements": (c_int, [_P, c_int, c_int, c_int, POINTER(c_double)]),
    "bdg_sw2d_curved_set_elements": (c_int, [_P, c_int, c_int, c_int, POINTER(c_double)]),
    "bdg_sw2d_curved_rk2_phase": (c_int, [_P, c_double, c_int, c_int]),
    "bdg_sw2d_curved_set_partition": (c_int, [_P, c_int, c_int, _P, c_int]),
    "bdg_sw2d_curved_comm_init": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int]),
    "bdg_sw2d_curved_step_rk2_exchanged": (c_int, [_P, c_double, c_int, c_int]),
    "bdg_sw2d_curved_lserk4_stages_exchanged": (c_int, [_P, c_double, c_int]),
    "bdg_sw2d_curved_exchange": (c_int, [_P, c_int]),
    "bdg_sw2d_curved_barrier": (c_int, [_P]),
}

#: every symbol include/blitzdg_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


def check(status):
    """Raise on a nonzero C-ABI status, with the library's message."""
    if status == BDG_OK:
        return
    msg = lib.bdg_last_error()
    msg = msg.decode("utf-8", "replace") if msg else "unknown error"
    if status == BDG_ERR_UNSTABLE:
        raise NumericalInstability(status, msg)
    if status == BDG_ERR_ARGUMENT:
        raise BdgError(status, "invalid argument: " + msg)
    raise BdgError(status, msg)


def table_to_numpy(tab, copy=True):
    """Materialise a borrowed bdg_table as an ndarray (fresh copy by default, like the
    reference's *_numpy() exporters, src/DGContext2D.cpp:24-176)."""
    n = tab.rows * tab.cols
    dtype = np.float64 if tab.dtype == BDG_F64 else np.int32
    if n == 0 or not tab.data:
        arr = np.zeros((tab.rows, tab.cols), dtype=dtype)
    else:
        ctype = c_double if tab.dtype == BDG_F64 else c_int
        buf = (ctype * n).from_address(tab.data)
        arr = np.frombuffer(buf, dtype=dtype).reshape(tab.rows, tab.cols)
    if tab.cols == 1:
        arr = arr.reshape(tab.rows)
    return arr.copy() if copy else arr


def as_f64(a, shape=None, name="array"):
    arr = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(arr.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(arr.shape)}")
    return arr


def as_i32(a, name="array"):
    return np.ascontiguousarray(a, dtype=np.int32)


def ptr(arr):
    return arr.ctypes.data_as(c_void_p) if arr is not None else None
