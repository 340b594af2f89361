"""ctypes binding of libvexhip.so (include/vexhip.h).

The product path has no fallback: if the HIP library is missing this module
raises, and every call that returns non-zero raises ``vexcl_amd.Error`` with
the library's ``file:line`` + HIP error text (the reference reports errors as
``vex::backend::error``, backend/cuda/error.hpp:119-145).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VEXHIP_LIBRARY") or os.path.join(_HERE, "lib", "libvexhip.so")   # VEXHIP_LIBRARY: A/B builds (tools/)
CSRC = os.path.join(_HERE, "csrc")


class Error(RuntimeError):
    """vex::error equivalent."""


def build(force=False, jobs=8):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(jobs)]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_vp = ctypes.c_void_p
c_f64 = ctypes.c_double
c_f32 = ctypes.c_float
c_u64 = ctypes.c_uint64
c_size = ctypes.c_size_t

F64, F32, I32, U32, I64, U64 = range(6)
SUM, SUM_KAHAN, MIN, MAX, MIN_MAX = range(5)


class DeviceProps(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 256), ("arch", ctypes.c_char * 64),
                ("compute_units", ctypes.c_int32), ("wavefront_size", ctypes.c_int32),
                ("max_threads_per_block", ctypes.c_int32), ("lds_bytes_per_block", ctypes.c_int32),
                ("clock_khz", ctypes.c_int32), ("l2_bytes", ctypes.c_int32),
                ("global_mem_bytes", ctypes.c_uint64), ("pci_bus_id", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class Traversal(ctypes.Structure):
    """vexhip_traversal (include/vexhip.h)."""
    _fields_ = [("grid_blocks", ctypes.c_int64), ("chunk", ctypes.c_int64), ("planes", ctypes.c_int64),
                ("plane_blocks", ctypes.c_int64), ("order", ctypes.c_void_p)]


class March(ctypes.Structure):
    """vexhip_march (include/vexhip.h)."""
    _fields_ = [("lo", ctypes.c_int32), ("hi", ctypes.c_int32), ("run", ctypes.c_int32), ("usable", ctypes.c_int32), ("x_last", ctypes.c_int64),
                ("nfar", ctypes.c_int32), ("far", ctypes.c_int32 * 3)]


class Plane(ctypes.Structure):
    """vexhip_plane (include/vexhip.h)."""
    _fields_ = [("usable", ctypes.c_int32), ("lines_per_plane", ctypes.c_int32), ("planes", ctypes.c_int32), ("depth", ctypes.c_int32),
                ("hot_block", ctypes.c_int32), ("tile", ctypes.c_int32), ("store_policy", ctypes.c_int32),
                ("table_pitch", ctypes.c_int32), ("flat", ctypes.c_int32), ("reserved", ctypes.c_int32), ("x_last", ctypes.c_int64)]


class Grid(ctypes.Structure):
    """vexhip_grid (include/vexhip.h)."""
    _fields_ = [("usable", ctypes.c_int32), ("nx", ctypes.c_int32), ("lines_per_plane", ctypes.c_int32), ("planes", ctypes.c_int32),
                ("depth", ctypes.c_int32), ("segments", ctypes.c_int32), ("segment_rows", ctypes.c_int32), ("threads", ctypes.c_int32),
                ("hot_class", ctypes.c_int32), ("classes", ctypes.c_int32), ("pitch", ctypes.c_int32), ("store_policy", ctypes.c_int32),
                ("flat", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("x_last", ctypes.c_int64), ("line_class", ctypes.c_void_p), ("table", ctypes.c_void_p)]


class SpMatInfo(ctypes.Structure):
    """vexhip_spmat_info (include/vexhip.h)."""
    _fields_ = [("format", ctypes.c_int32), ("value_type", ctypes.c_int32), ("device", ctypes.c_int32),
                ("ndeltas", ctypes.c_int32), ("nvalues", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("rows", ctypes.c_int64), ("nnz", ctypes.c_int64), ("ell_width", ctypes.c_int64),
                ("tail_nnz", ctypes.c_int64), ("sell_bytes", ctypes.c_int64), ("matrix_bytes", ctypes.c_int64),
                ("sell", ctypes.c_void_p), ("deltas", ctypes.c_void_p), ("values", ctypes.c_void_p),
                ("csr_ptr", ctypes.c_void_p), ("csr_col", ctypes.c_void_p), ("csr_val", ctypes.c_void_p),
                ("traversal", Traversal), ("slice_blocks", ctypes.c_void_p), ("code_pool", ctypes.c_void_p),
                ("dictionary_blocks", ctypes.c_int64), ("march", March), ("plane", Plane), ("grid", Grid),
                ("product", ctypes.c_char * 64), ("reason", ctypes.c_char * 320)]


SPMAT_AUTO, SPMAT_SELL8V, SPMAT_SELL8, SPMAT_SELL, SPMAT_CSR = range(5)
SPMAT_BORROW_CSR = 1
SPMAT_NO_DICTIONARY = 2
SPMAT_NO_MARCH = 4
SPMAT_NO_PLANE = 8
SPMAT_NO_GRID_BUILD = 16
SPMAT_SQUARE = 32
SPMAT_PLAIN_ORDER = 64
SPMAT_NAMES = {SPMAT_SELL8V: "sell8v", SPMAT_SELL8: "sell8", SPMAT_SELL: "sell32", SPMAT_CSR: "csr"}

# name -> (restype, argtypes); restype None means "int status, checked"
_PROTOS = {
    "vexhip_last_error": (ctypes.c_char_p, []),
    "vexhip_last_error_code": (ctypes.c_int, []),
    "vexhip_abi_version": (c_int, []),
    "vexhip_device_count": (None, [ctypes.POINTER(c_int)]),
    "vexhip_device_get_props": (None, [c_int, ctypes.POINTER(DeviceProps)]),
    "vexhip_device_sync": (None, [c_int]),
    "vexhip_mem_info": (None, [c_int, ctypes.POINTER(c_u64), ctypes.POINTER(c_u64)]),
    "vexhip_stream_create": (None, [c_int, ctypes.POINTER(c_vp)]),
    "vexhip_stream_destroy": (None, [c_int, c_vp]),
    "vexhip_stream_sync": (None, [c_int, c_vp]),
    "vexhip_event_create": (None, [c_int, c_int, ctypes.POINTER(c_vp)]),
    "vexhip_event_destroy": (None, [c_int, c_vp]),
    "vexhip_event_record": (None, [c_int, c_vp, c_vp]),
    "vexhip_event_sync": (None, [c_int, c_vp]),
    "vexhip_stream_wait_event": (None, [c_int, c_vp, c_vp]),
    "vexhip_event_elapsed_ms": (None, [c_int, c_vp, c_vp, ctypes.POINTER(c_f32)]),
    "vexhip_reload_env": (None, []),
    "vexhip_malloc": (None, [c_int, c_size, ctypes.POINTER(c_vp)]),
    "vexhip_malloc_placement": (c_size, [c_size, ctypes.c_uint64, ctypes.c_uint]),
    "vexhip_malloc_stagger": (c_size, [c_size, ctypes.c_uint]),
    "vexhip_malloc_managed": (None, [c_int, c_size, ctypes.POINTER(c_vp)]),
    "vexhip_free": (None, [c_int, c_vp]),
    "vexhip_memcpy_h2d": (None, [c_int, c_vp, c_vp, c_size, c_vp, c_int]),
    "vexhip_memcpy_d2h": (None, [c_int, c_vp, c_vp, c_size, c_vp, c_int]),
    "vexhip_memcpy_d2d": (None, [c_int, c_vp, c_vp, c_size, c_vp]),
    "vexhip_memcpy_peer": (None, [c_int, c_vp, c_int, c_vp, c_size, c_vp]),
    "vexhip_memset": (None, [c_int, c_vp, c_int, c_size, c_vp]),
    "vexhip_host_alloc": (None, [c_size, ctypes.POINTER(c_vp)]),
    "vexhip_host_free": (None, [c_vp]),
    "vexhip_module_compile": (None, [c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(c_vp)]),
    "vexhip_module_unload": (None, [c_int, c_vp]),
    "vexhip_module_get_function": (None, [c_int, c_vp, ctypes.c_char_p, ctypes.POINTER(c_vp)]),
    "vexhip_function_max_threads": (None, [c_int, c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "vexhip_launch": (None, [c_int, c_vp] + [ctypes.c_uint] * 7 + [c_vp, ctypes.POINTER(c_vp)]),
    "vexhip_jit_stats": (None, [ctypes.POINTER(c_u64), ctypes.POINTER(c_u64)]),
    "vexhip_jit_check": (None, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]),
    "vexhip_spmv_csr_f64_i32": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vexhip_spmv_csr_f32_i32": (None, [c_int, c_vp, c_i64, c_f32, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vexhip_spmv_csr_f64_i64": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vexhip_spmv_csr_rows_f64_i32": (None, [c_int, c_vp, c_i64, c_f64] + [c_vp] * 6),
    "vexhip_spmv_csr_rows_f32_i32": (None, [c_int, c_vp, c_i64, c_f32] + [c_vp] * 6),
    "vexhip_spmv_csr_set_variant": (None, [c_int]),
    "vexhip_csr_traversal_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_int, ctypes.POINTER(Traversal)]),
    "vexhip_spmv_csr_ordered_f64_i32": (None, [c_int, c_vp, c_i64, c_f64, c_int] + [c_vp] * 5 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmv_csr_ordered_f32_i32": (None, [c_int, c_vp, c_i64, c_f32, c_int] + [c_vp] * 5 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmv_hell_set_variant": (None, [c_int]),
    "vexhip_spmv_hell_f64_i32": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_i64, c_i64] + [c_vp] * 7),
    "vexhip_spmv_hell_f32_i32": (None, [c_int, c_vp, c_i64, c_f32, c_int, c_i64, c_i64] + [c_vp] * 7),
    "vexhip_hell_order_capacity": (c_i64, [c_i64]),
    "vexhip_hell_order_i32": (None, [c_int, c_vp, c_i64, c_i64, c_i64, c_vp, c_int, c_vp, c_i64, ctypes.POINTER(Traversal)]),
    "vexhip_sell_order_i32": (None, [c_int, c_vp, c_i64, c_i64, c_int, c_vp, c_int, c_vp, c_i64, ctypes.POINTER(Traversal)]),
    "vexhip_spmv_hell_ordered_f64_i32": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_i64, c_i64] + [c_vp] * 7 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmv_hell_ordered_f32_i32": (None, [c_int, c_vp, c_i64, c_f32, c_int, c_i64, c_i64] + [c_vp] * 7 + [ctypes.POINTER(Traversal)]),
    "vexhip_sell_bytes": (c_i64, [c_i64, c_i64, c_int]),
    "vexhip_sell_fill_f64_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "vexhip_sell_fill_f32_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "vexhip_spmv_sell_f64_i32": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_i64] + [c_vp] * 6 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmv_sell_f32_i32": (None, [c_int, c_vp, c_i64, c_f32, c_int, c_i64] + [c_vp] * 6 + [ctypes.POINTER(Traversal)]),
    "vexhip_sell8_bytes": (c_i64, [c_i64, c_i64, c_int]),
    "vexhip_sell8_analyze_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, ctypes.POINTER(c_int)]),
    "vexhip_sell8_fill_f64_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_vp, ctypes.POINTER(Traversal)]),
    "vexhip_sell8_fill_f32_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_vp, ctypes.POINTER(Traversal)]),
    "vexhip_spmv_sell8_f64_i32": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_i64] + [c_vp] * 7 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmv_sell8_f32_i32": (None, [c_int, c_vp, c_i64, c_f32, c_int, c_i64] + [c_vp] * 7 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmv_sell8_set_variant": (None, [c_int]),
    "vexhip_sell8v_bytes": (c_i64, [c_i64, c_i64]),
    "vexhip_sell8v_analyze_f64_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, ctypes.POINTER(c_int)]),
    "vexhip_sell8v_analyze_f32_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, ctypes.POINTER(c_int)]),
    "vexhip_sell8v_fill_f64_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_vp, c_int, c_vp, ctypes.POINTER(Traversal)]),
    "vexhip_sell8v_fill_f32_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_vp, c_int, c_vp, ctypes.POINTER(Traversal)]),
    "vexhip_spmv_sell8v_f64_i32": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_i64] + [c_vp] * 8 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmv_sell8v_dict_f64_i32": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_i64] + [c_vp] * 9 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmv_sell8v_dict_f32_i32": (None, [c_int, c_vp, c_i64, c_f32, c_int, c_i64] + [c_vp] * 9 + [ctypes.POINTER(Traversal)]),
    "vexhip_slice_dictionary": (None, [c_int, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, ctypes.POINTER(c_i64)]),
    "vexhip_spmv_sell8_dict_f64_i32": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_i64] + [c_vp] * 9 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmv_sell8_dict_f32_i32": (None, [c_int, c_vp, c_i64, c_f32, c_int, c_i64] + [c_vp] * 9 + [ctypes.POINTER(Traversal)]),
    "vexhip_sell8_march_plan": (None, [c_int, c_vp, c_vp, c_int, c_vp, c_i64, c_int, ctypes.POINTER(Traversal), c_i64, ctypes.POINTER(March)]),
    "vexhip_sell8_last_fill_max_col": (c_i64, []),
    "vexhip_stream_copy_f64": (None, [c_int, c_vp, c_vp, c_vp, c_i64]),
    "vexhip_sell8_plane_plan": (None, [c_int, c_vp, c_vp, c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_i64, ctypes.POINTER(Plane)]),
    "vexhip_sell8_grid_plan": (None, [c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_i64, ctypes.POINTER(Grid)]),
    "vexhip_sell8_grid_release": (None, [c_int, ctypes.POINTER(Grid)]),
    "vexhip_sell8_grid_geometry": (None, [c_int, c_i64, c_i64, c_i64, ctypes.POINTER(Grid)]),
    "vexhip_sell8_grid_virtual_line": (c_i64, [c_i64]),
    "vexhip_sell8_plane_geometry": (None, [c_int, c_i64, c_i64, ctypes.POINTER(Plane)]),
    "vexhip_sell8_plane_f32_depth": (c_i64, [c_int, c_i64, c_i64]),
    "vexhip_sell8_grid_check": (None, [ctypes.POINTER(Grid), c_i64]),
    "vexhip_spmv_sell8v_grid_f64": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_vp, c_vp, c_vp, ctypes.POINTER(Grid)]),
    "vexhip_spmv_sell8v_grid_f32": (None, [c_int, c_vp, c_i64, c_f32, c_int, c_vp, c_vp, c_vp, ctypes.POINTER(Grid)]),
    "vexhip_spmv_sell8v_plane_f64_i32": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_i64] + [c_vp] * 6 + [ctypes.POINTER(Plane)]),
    "vexhip_spmv_sell8v_plane_f32_i32": (None, [c_int, c_vp, c_i64, ctypes.c_float, c_int, c_i64] + [c_vp] * 6 + [ctypes.POINTER(Plane)]),
    "vexhip_spmv_sell8v_march_f64_i32": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_i64] + [c_vp] * 9 + [ctypes.POINTER(Traversal), ctypes.POINTER(March)]),
    "vexhip_spmv_sell8v_march_f32_i32": (None, [c_int, c_vp, c_i64, c_f32, c_int, c_i64] + [c_vp] * 9 + [ctypes.POINTER(Traversal), ctypes.POINTER(March)]),
    "vexhip_spmm_sell8_dict_f64_i32": (None, [c_int, c_vp, c_i64, c_int, c_f64, c_int, c_i64] + [c_vp] * 9 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmm_sell8_dict_f32_i32": (None, [c_int, c_vp, c_i64, c_int, c_f32, c_int, c_i64] + [c_vp] * 9 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmv_sell8v_f32_i32": (None, [c_int, c_vp, c_i64, c_f32, c_int, c_i64] + [c_vp] * 8 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmat_create_f64_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_int, ctypes.POINTER(c_vp)]),
    "vexhip_spmat_create_f32_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_int, ctypes.POINTER(c_vp)]),
    "vexhip_spmat_create_f64_p64": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_int, ctypes.POINTER(c_vp)]),
    "vexhip_spmat_create_f32_p64": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_int, ctypes.POINTER(c_vp)]),
    "vexhip_spmat_destroy": (None, [c_vp]),
    "vexhip_spmat_apply_f64": (None, [c_vp, c_vp, c_f64, c_int, c_vp, c_vp]),
    "vexhip_spmat_apply_axpby_f64": (None, [c_vp, c_vp, c_f64, c_vp, c_f64, c_vp, c_vp]),
    "vexhip_spmat_axpby_fused": (c_int, [c_vp, c_vp, c_vp, c_vp]),
    "vexhip_spmat_apply_axpby_f32": (None, [c_vp, c_vp, c_f32, c_vp, c_f32, c_vp, c_vp]),
    "vexhip_spmat_apply_f32": (None, [c_vp, c_vp, c_f32, c_int, c_vp, c_vp]),
    "vexhip_spmat_apply_multi_f64": (None, [c_vp, c_vp, c_int, c_f64, c_int, c_vp, c_vp]),
    "vexhip_spmat_apply_multi_f32": (None, [c_vp, c_vp, c_int, c_f32, c_int, c_vp, c_vp]),
    "vexhip_spmat_get_info": (None, [c_vp, ctypes.POINTER(SpMatInfo)]),
    "vexhip_csr_split_sizes_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, ctypes.POINTER(c_i64)]),
    "vexhip_csr_split_f64_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, ctypes.POINTER(c_i64)] + [c_vp] * 8),
    "vexhip_csr_split_f32_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, ctypes.POINTER(c_i64)] + [c_vp] * 8),
    "vexhip_comm_unique_id": (None, [c_vp]),
    "vexhip_comm_init": (None, [c_int, ctypes.POINTER(c_int), c_int, ctypes.POINTER(c_vp)]),
    "vexhip_comm_init_rank": (None, [c_int, c_int, c_int, c_vp, ctypes.POINTER(c_vp)]),
    "vexhip_comm_destroy": (None, [c_vp]),
    "vexhip_comm_size": (None, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "vexhip_halo_exchange": (None, [c_vp, c_int, ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), ctypes.POINTER(c_vp)]),
    "vexhip_allreduce_scalar": (None, [c_vp, c_int, c_int, ctypes.POINTER(c_vp), c_i64, ctypes.POINTER(c_vp)]),
    "vexhip_allgather": (None, [c_vp, c_int, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), c_i64, ctypes.POINTER(c_vp)]),
    "vexhip_dist_spmv_create": (None, [c_vp, c_int, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, ctypes.POINTER(c_i64),
                                       c_i64, c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_vp)]),
    "vexhip_dist_spmv_destroy": (None, [c_vp]),
    "vexhip_dist_spmv_set_graph": (None, [c_vp, c_int]),
    "vexhip_dist_spmv_apply": (None, [c_vp, c_vp, c_f64, c_int, c_vp, c_vp]),
    "vexhip_comm_rccl_info": (None, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "vexhip_ipc_window_create": (None, [c_int, c_int, c_int, c_i64, ctypes.POINTER(c_vp)]),
    "vexhip_ipc_window_export": (None, [c_vp, c_vp]),
    "vexhip_ipc_window_open": (None, [c_vp, c_int, c_vp]),
    "vexhip_ipc_window_data": (None, [c_vp, ctypes.POINTER(c_vp)]),
    "vexhip_ipc_window_destroy": (None, [c_vp]),
    "vexhip_dist_spmv_debug": (None, [c_vp, c_vp, c_i64]),
    "vexhip_dist_spmv_create_halo": (None, [c_vp, c_vp, c_i64, c_i64, c_int, c_int, ctypes.POINTER(c_vp)]),
    "vexhip_dist_spmv_create_halo_pull": (None, [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, ctypes.POINTER(c_vp)]),
    "vexhip_dist_spmv_apply_pull": (None, [c_vp, c_vp, c_f64, c_int, c_vp, c_vp, c_vp, c_vp]),
    "vexhip_ipc_window_attach": (None, [c_vp, c_int, c_vp]),
    "vexhip_ipc_export": (None, [c_int, c_vp, c_vp, ctypes.POINTER(c_i64)]),
    "vexhip_ipc_open": (None, [c_int, c_vp, ctypes.POINTER(c_vp)]),
    "vexhip_ipc_close": (None, [c_int, c_vp]),
    "vexhip_csr_extend_halo_i32": (None, [c_int, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, ctypes.POINTER(c_i64)]),
    "vexhip_dist_spmv_create_ipc": (None, [c_vp, c_int, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, ctypes.POINTER(c_i64),
                                           ctypes.POINTER(c_i64), c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_vp)]),
    "vexhip_dist_spmv_status": (None, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "vexhip_dist_spmv_profile": (None, [c_vp, c_vp, c_f64, c_int, c_vp, c_vp, ctypes.POINTER(c_f32)]),
    "vexhip_spmm_sell8_f64_i32": (None, [c_int, c_vp, c_i64, c_int, c_f64, c_int, c_i64] + [c_vp] * 7 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmm_sell8_f32_i32": (None, [c_int, c_vp, c_i64, c_int, c_f32, c_int, c_i64] + [c_vp] * 7 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmm_sell8v_f64_i32": (None, [c_int, c_vp, c_i64, c_int, c_f64, c_int, c_i64] + [c_vp] * 8 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmm_sell8v_dict_f64_i32": (None, [c_int, c_vp, c_i64, c_int, c_f64, c_int, c_i64] + [c_vp] * 9 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmm_sell8v_dict_f32_i32": (None, [c_int, c_vp, c_i64, c_int, c_f32, c_int, c_i64] + [c_vp] * 9 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmm_sell8v_f32_i32": (None, [c_int, c_vp, c_i64, c_int, c_f32, c_int, c_i64] + [c_vp] * 8 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmm_sell_f64_i32": (None, [c_int, c_vp, c_i64, c_int, c_f64, c_int, c_i64] + [c_vp] * 6 + [ctypes.POINTER(Traversal)]),
    "vexhip_spmm_sell_f32_i32": (None, [c_int, c_vp, c_i64, c_int, c_f32, c_int, c_i64] + [c_vp] * 6 + [ctypes.POINTER(Traversal)]),
    "vexhip_hell_analyze_i32": (None, [c_int, c_vp, c_i64, c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "vexhip_hell_fill_f64_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64] + [c_vp] * 5),
    "vexhip_hell_fill_f32_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64] + [c_vp] * 5),
    "vexhip_gather_f64_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "vexhip_gather_f32_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "vexhip_reduce_tmp_bytes": (c_size, []),
    "vexhip_reduce": (None, [c_int, c_vp, c_int, c_int, c_vp, c_i64, c_vp, c_vp]),
    "vexhip_reduce_dot": (None, [c_int, c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "vexhip_reduce_finish": (None, [c_int, c_vp, c_int, c_int, c_vp, c_i64, c_vp]),
    "vexhip_reduce_num_groups": (None, [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "vexhip_scan_set_lookback": (None, [c_int]),
    "vexhip_scan_tmp_bytes": (c_size, [c_int, c_i64]),
    "vexhip_scan": (None, [c_int, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "vexhip_sort_tmp_bytes": (c_size, [c_int, c_i64]),
    "vexhip_sort_set_rank": (None, [c_int]),
    "vexhip_sort_status": (None, [c_int, c_vp, c_i64, c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "vexhip_sort": (None, [c_int, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_vp]),
    "vexhip_spmv_ccsr_f64": (None, [c_int, c_vp, c_i64, c_f64, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "vexhip_spmv_ccsr_set_rows_per_lane": (None, [c_int]),
    "vexhip_ccsr_to_csr_f64_i32": (None, [c_int, c_vp, c_i64] + [c_vp] * 7 + [ctypes.POINTER(c_i64)]),
    "vexhip_ccsr_to_csr_f32_i32": (None, [c_int, c_vp, c_i64] + [c_vp] * 7 + [ctypes.POINTER(c_i64)]),
    "vexhip_spmv_ccsr_f32": (None, [c_int, c_vp, c_i64, c_f32, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "vexhip_stencil_conv_f64": (None, [c_int, c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_f64, c_f64]),
    "vexhip_stencil_conv_f32": (None, [c_int, c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32]),
    "vexhip_poisson3d_nnz": (c_i64, [c_i64]),
    "vexhip_poisson3d_strip_nnz": (c_i64, [c_i64, c_i64, c_i64]),
    "vexhip_poisson3d_csr_f64_i32": (None, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "vexhip_poisson3d_strip_f64_i32": (None, [c_int, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "vexhip_poisson3d_strip_f64_p64": (None, [c_int, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "vexhip_diffusion3d_strip_f64_i32": (None, [c_int, c_vp, c_i64, c_i64, c_i64, c_u64, c_vp, c_vp, c_vp]),
    "vexhip_fill_hash": (None, [c_int, c_vp, c_int, c_u64, c_vp, c_i64]),
    "vexhip_fill_value": (None, [c_int, c_vp, c_int, c_vp, c_vp, c_i64]),
    "vexhip_mba_fit": (None, [c_int, c_vp, c_int, c_int, ctypes.POINTER(c_f64), ctypes.POINTER(c_f64), c_vp, c_vp, c_i64,
                       ctypes.POINTER(c_size), c_int, c_f64, ctypes.POINTER(c_f64), ctypes.POINTER(c_f64),
                       ctypes.POINTER(c_size), ctypes.POINTER(c_size), ctypes.POINTER(c_vp), ctypes.POINTER(c_size)]),
    "vexhip_fft_best_size": (c_size, [c_size]),
    "vexhip_fft_plan_create": (None, [c_int, c_int, c_int, ctypes.POINTER(c_size), ctypes.POINTER(c_int), ctypes.POINTER(c_vp)]),
    "vexhip_fft_plan_destroy": (None, [c_vp]),
    "vexhip_fft_exec": (None, [c_vp, c_vp, c_vp, c_vp]),
    "vexhip_fft_plan_steps": (None, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
}

EXPORTS = tuple(sorted(_PROTOS))


class _Lib:
    def __init__(self, path):
        if not os.path.exists(path):
            raise Error(
                "libvexhip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C vexcl_amd/csrc`.  There is no CPU fallback." % path)
        self.path = path
        self.cdll = ctypes.CDLL(path)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(self.cdll, name)        # AttributeError = missing export
            fn.argtypes = args
            fn.restype = c_int if res is None else res
            setattr(self, name[len("vexhip_"):], self._checked(fn) if res is None else fn)

    def _checked(self, fn):
        last_error = self.cdll.vexhip_last_error

        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise Error(last_error().decode(errors="replace"))
        call.__name__ = fn.__name__
        return call


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib(LIB_PATH)
    return _lib
