"""ctypes bindings of libzvdb_hip.so (include/zvdb_hip.h) and libzvdb_host.so.

The product path has NO CPU fallback: if the shared library is missing this module raises, and
every entry point that needs a GPU fails with ZVDB_ERR_NO_DEVICE when none is visible."""
from __future__ import annotations

import ctypes as C
import os

from . import _build

OK, ERR_INVALID, ERR_DIM_MISMATCH, ERR_OOM, ERR_NODE_NOT_FOUND, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_DEVICE = range(8)
METRIC_L2SQ, METRIC_DOT, METRIC_COSINE = 0, 1, 2
BF_F32, BF_BF16, BF_I8, BF_FP4 = 0, 1, 2, 3
FLAG_BF16_MIRROR = 1
FLAG_BF16_ROWS = 2
ENTRY_DESCEND = 0xFFFFFFFFFFFFFFFF  # search_graph entry: hierarchical descent from the bulk graph's top node
ABI_VERSION = 1
COMM_ID_BYTES = 128
SHARDED_GRAPH = 0xFFFFFFFF


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("dim", C.c_uint32), ("metric", C.c_uint32),
                ("m", C.c_uint32), ("ef_construction", C.c_uint32), ("device", C.c_int32),
                ("flags", C.c_uint32), ("graph_stride", C.c_uint32), ("capacity", C.c_uint64)]


class SearchStats(C.Structure):
    _fields_ = [("dist_evals", C.c_uint64), ("pops", C.c_uint64)]


class CertInfo(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("batches", C.c_uint64), ("repaired", C.c_uint64),
                ("level", C.c_uint32), ("enabled", C.c_uint32), ("rerank_rows", C.c_uint64),
                ("rerank_cycles", C.c_uint32 * 5), ("guess_misses", C.c_uint32),
                ("bound_checked", C.c_uint32), ("bound_violations", C.c_uint32)]


OPT_CERTIFY, OPT_CERT_MIN_LEVEL, OPT_CERT_FORCE_REPAIR, OPT_EF_SEARCH_WIDTH = 1, 2, 3, 4


# every symbol include/zvdb_hip.h declares: (name, restype, argtypes)
_vp, _u64, _u32, _i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
_f32p, _u64p, _u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
HIP_SYMBOLS = [
    ("zvdb_hip_last_error", C.c_char_p, []),
    ("zvdb_hip_abi_version", _i32, []),
    ("zvdb_hip_device_count", _i32, [C.POINTER(C.c_int)]),
    ("zvdb_hip_create", _i32, [C.POINTER(Config), C.POINTER(_vp)]),
    ("zvdb_hip_destroy", None, [_vp]),
    ("zvdb_hip_count", _i32, [_vp, _u64p]),
    ("zvdb_hip_dim", _i32, [_vp, _u32p]),
    ("zvdb_hip_add", _i32, [_vp, _f32p, _u64, _u32, _u64p]),
    ("zvdb_hip_add_dev", _i32, [_vp, _vp, _u64, _u32, _vp, _u64p]),
    ("zvdb_hip_add_synthetic", _i32, [_vp, _u64, _u32, _u64, _u64, _u64p]),
    ("zvdb_hip_fill_synthetic_dev", _i32, [_vp, _u64, _u32, _u64, _u64, _vp]),
    ("zvdb_hip_get_vector", _i32, [_vp, _u64, _f32p, _u32]),
    ("zvdb_hip_set_neighbors", _i32, [_vp, _u64, _u64p, _u32]),
    ("zvdb_hip_upload_graph", _i32, [_vp, _u32p, _u32p, _u64, _u32]),
    ("zvdb_hip_get_neighbors", _i32, [_vp, _u64, _u64p, _u32, _u32p]),
    ("zvdb_hip_build_knn_graph", _i32, [_vp, _u32, _u32]),
    ("zvdb_hip_distances", _i32, [_vp, _f32p, _u32, _u64p, _u32, _f32p]),
    ("zvdb_hip_distances_from", _i32, [_vp, _u64, _u64p, _u32, _f32p]),
    ("zvdb_hip_greedy_layer0", _i32, [_vp, _u64, _u64p, _f32p, _u64p]),
    ("zvdb_hip_insert_batch", _i32, [_vp, _f32p, _u32, _u64, _u32p, _u64p, _u32p, _u32p, _u64]),
    ("zvdb_hip_insert_batch_layer0", _i32, [_vp, _f32p, _u32, _u64, _u64, _u64p, _u64p, _f32p, _u64p, _u32p, _u32p]),
    ("zvdb_hip_insert_layer0", _i32, [_vp, _f32p, _u32, _u64, _u64p, _u64p, _f32p, _u64p, _u32p, _u32p, _u64p]),
    ("zvdb_hip_search_graph", _i32, [_vp, _f32p, _u32, _u32, _u32, _u32, _u64, _u64p, _f32p, _u32p,
                                      C.POINTER(SearchStats)]),
    ("zvdb_hip_search_graph_reserve", _i32, [_vp, _u32, _u32, _u32]),
    ("zvdb_hip_search_bruteforce", _i32, [_vp, _f32p, _u32, _u32, _u32, _u32, _u64p, _f32p, _u32p]),
    ("zvdb_hip_bruteforce_workspace", _i32, [_vp, _u32, _u32, _u32, _u64p]),
    ("zvdb_hip_search_bruteforce_dev", _i32, [_vp, _vp, _u32, _u32, _u32, _u32, _u64, _vp, _u64, _vp, _vp,
                                              _vp, _vp]),
    ("zvdb_hip_search_graph_dev", _i32, [_vp, _vp, _u32, _u32, _u32, _u32, _u64, _u64, _vp, _vp, _vp, _vp,
                                         _vp]),
    ("zvdb_hip_build_graph", _i32, [_vp, _u32, _u32, _u32, _u64]),
    ("zvdb_hip_graph_info", _i32, [_vp, _u64p, _u32p]),
    ("zvdb_hip_storage_info", _i32, [_vp, _u64p, _u64p]),
    ("zvdb_hip_save", _i32, [_vp, C.c_char_p]),
    ("zvdb_hip_load", _i32, [C.c_char_p, _i32, C.POINTER(_vp)]),
    ("zvdb_hip_get_neighbors_layer", _i32, [_vp, _u64, _u32, _u64p, _u32, _u32p]),
    ("zvdb_hip_merge_topk_dev", _i32, [_vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    ("zvdb_hip_merge_topk_packed_dev", _i32, [_vp, _u64, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    ("zvdb_hip_index_set_shard_count", _i32, [_vp, _u32]),
    ("zvdb_hip_index_set_screen_survivors", _i32, [_vp, _u32]),
    ("zvdb_hip_index_set_option", _i32, [_vp, _u32, _u64]),
    ("zvdb_hip_set_tuning", _i32, [C.c_char_p, C.c_char_p]),
    ("zvdb_hip_get_tuning", C.c_char_p, [C.c_char_p]),
    ("zvdb_hip_bruteforce_cert_info", _i32, [_vp, C.POINTER(CertInfo)]),
    ("zvdb_hip_comm_unique_id", _i32, [_vp]),
    ("zvdb_hip_comm_create", _i32, [_u32, _u32, _vp, C.c_int32, C.POINTER(_vp)]),
    ("zvdb_hip_comm_destroy", None, [_vp]),
    ("zvdb_hip_comm_info", _i32, [_vp, _u32p, _u32p]),
    ("zvdb_hip_comm_precheck", _i32, [C.c_int32]),
    ("zvdb_hip_comm_query", _i32, [_vp, _u32p, _u32p, C.c_char_p, _u32]),
    ("zvdb_hip_comm_timing_enable", _i32, [_vp, _i32]),
    ("zvdb_hip_comm_timing_read", _i32, [_vp, C.POINTER(C.c_double), _u32p]),
    ("zvdb_hip_sharded_workspace", _i32, [_vp, _vp, _u32, _u32, _u32, _u64p]),
    ("zvdb_hip_search_bruteforce_sharded_dev", _i32, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _u64, _vp, _u64, _vp, _vp, _vp, _vp]),
    ("zvdb_hip_search_graph_sharded_dev", _i32, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _u64, _u64, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    ("zvdb_hip_timing_enable", _i32, [_vp, _i32]),
    ("zvdb_hip_timing_read", _i32, [_vp, C.POINTER(C.c_double), _u32p]),
]

HOST_SYMBOLS = [
    ("zvdb_host_last_error", C.c_char_p, []),
    ("zvdb_host_hnsw_init", _vp, [C.c_size_t, C.c_size_t, _i32, _u32]),
    ("zvdb_host_hnsw_deinit", None, [_vp]),
    ("zvdb_host_hnsw_insert", _i32, [_vp, _f32p, C.c_size_t, _i32]),
    ("zvdb_host_hnsw_insert_many", _i32, [_vp, _f32p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]),
    ("zvdb_host_hnsw_search", _i32, [_vp, _f32p, C.c_size_t, C.c_size_t, _u64p, _f32p,
                                      C.POINTER(C.c_size_t), _u64p, _u64p]),
    ("zvdb_host_hnsw_search_many", _i32, [_vp, _f32p, C.c_size_t, C.c_size_t, C.c_size_t, _u64p, _f32p, _u32p]),
    ("zvdb_host_hnsw_count", C.c_size_t, [_vp]),
    ("zvdb_host_hnsw_dim", C.c_size_t, [_vp]),
    ("zvdb_host_hnsw_max_level", C.c_size_t, [_vp]),
    ("zvdb_host_hnsw_entry_point", C.c_longlong, [_vp]),
    ("zvdb_host_hnsw_handle", _vp, [_vp]),
    ("zvdb_host_hnsw_connections", C.c_longlong, [_vp, C.c_size_t, C.c_size_t, _u64p, C.c_size_t]),
]

_hip = None
_host = None


def _bind(lib, table):
    for name, res, args in table:
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def hip():
    """libzvdb_hip.so, bound.  Raises if the HIP extension has not been built."""
    global _hip
    if _hip is None:
        if not os.path.exists(_build.LIB_HIP):
            raise RuntimeError(
                "libzvdb_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C zvdb_amd/csrc`); zvdb_amd has no CPU fallback")
        _hip = _bind(C.CDLL(_build.LIB_HIP, mode=C.RTLD_GLOBAL), HIP_SYMBOLS)
        if _hip.zvdb_hip_abi_version() != ABI_VERSION:
            raise RuntimeError("libzvdb_hip.so ABI version mismatch")
    return _hip


def host():
    global _host
    if _host is None:
        hip()
        if not os.path.exists(_build.LIB_HOST):
            raise RuntimeError("libzvdb_host.so is not built; zvdb_amd has no CPU fallback")
        _host = _bind(C.CDLL(_build.LIB_HOST), HOST_SYMBOLS)
    return _host


class ZvdbError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"zvdb_hip error {code}: {msg}")
        self.code = code


class ZvdbPanic(Exception):
    """The reference @panic's (hnsw.zig:183-185) -- not a recoverable error."""


def check(rc: int) -> None:
    if rc == OK:
        return
    msg = hip().zvdb_hip_last_error().decode(errors="replace")
    if rc == ERR_DIM_MISMATCH:
        raise ZvdbPanic(msg)
    raise ZvdbError(rc, msg)


def set_tuning(name: str, value=None) -> None:
    """zvdb_hip_set_tuning: an A/B / test switch of the launch code (DESIGN.md "Runtime switches"); value None removes it."""
    check(hip().zvdb_hip_set_tuning(name.encode(), None if value is None else str(value).encode()))


class tuned:
    """with tuned(ZVDB_EF_BEAM="lds", ...): the switches are set inside the block and removed after it."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        for k, v in self.kv.items():
            set_tuning(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kv:
            set_tuning(k, None)
        return False


def device_count() -> int:
    n = C.c_int(0)
    hip().zvdb_hip_device_count(C.byref(n))
    return n.value
