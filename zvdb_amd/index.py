"""`Index` -- thin ctypes view of one `zvdb_hip_index` (include/zvdb_hip.h): the device-resident
shard that the kernels work on.  Host-pointer calls take numpy arrays; `_dev` calls take raw device
pointers (e.g. ``torch.Tensor.data_ptr()``) plus a stream handle and only enqueue work."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import Config, SearchStats, check

_f32p, _u64p, _u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Index:
    def __init__(self, dim: int = 0, metric: int = _lib.METRIC_L2SQ, m: int = 16, ef_construction: int = 200,
                 device: int = -1, flags: int = 0, graph_stride: int = 0, capacity: int = 0):
        self._L = _lib.hip()
        cfg = Config(_lib.ABI_VERSION, dim, metric, m, ef_construction, device, flags, graph_stride, capacity)
        h = C.c_void_p()
        check(self._L.zvdb_hip_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._own = True

    @classmethod
    def _from_handle(cls, handle) -> "Index":
        self = cls.__new__(cls)
        self._L = _lib.hip()
        self._h = C.c_void_p(handle)
        self._own = False
        return self

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h and self._own:
            self._L.zvdb_hip_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- rows ------------------------------------------------------------------
    def count(self) -> int:
        n = C.c_uint64(0)
        check(self._L.zvdb_hip_count(self._h, C.byref(n)))
        return n.value

    @property
    def dim(self) -> int:
        d = C.c_uint32(0)
        check(self._L.zvdb_hip_dim(self._h, C.byref(d)))
        return d.value

    def add(self, rows) -> int:
        r = _f32(rows)
        if r.ndim == 1:
            r = r.reshape(1, -1)
        first = C.c_uint64(0)
        check(self._L.zvdb_hip_add(self._h, r.ctypes.data_as(_f32p), r.shape[0], r.shape[1], C.byref(first)))
        return first.value

    def add_dev(self, ptr: int, n: int, dim: int, stream: int = 0) -> int:
        first = C.c_uint64(0)
        check(self._L.zvdb_hip_add_dev(self._h, ptr, n, dim, stream, C.byref(first)))
        return first.value

    def add_synthetic(self, n: int, dim: int, seed: int, first_row: int = 0) -> int:
        first = C.c_uint64(0)
        check(self._L.zvdb_hip_add_synthetic(self._h, n, dim, seed, first_row, C.byref(first)))
        return first.value

    def get_vector(self, i: int) -> np.ndarray:
        out = np.zeros(self.dim, dtype=np.float32)
        check(self._L.zvdb_hip_get_vector(self._h, i, out.ctypes.data_as(_f32p), out.size))
        return out

    # ---- graph mirror ------------------------------------------------------------
    def set_neighbors(self, node: int, ids) -> None:
        a = np.ascontiguousarray(ids, dtype=np.uint64)
        check(self._L.zvdb_hip_set_neighbors(self._h, node, a.ctypes.data_as(_u64p), a.size))

    def upload_graph(self, adj, deg) -> None:
        adj = np.ascontiguousarray(adj, dtype=np.uint32)
        deg = np.ascontiguousarray(deg, dtype=np.uint32)
        check(self._L.zvdb_hip_upload_graph(self._h, adj.ctypes.data_as(_u32p), deg.ctypes.data_as(_u32p),
                                            adj.shape[0], adj.shape[1]))

    def build_knn_graph(self, degree: int, mode: int = _lib.BF_F32) -> None:
        """Batched GPU construction of the layer-0 graph: exact `degree`-NN lists."""
        check(self._L.zvdb_hip_build_knn_graph(self._h, degree, mode))

    def build_graph(self, M: int, ef_construction: int = 64, mode: int = _lib.BF_I8, level_seed: int = 1) -> None:
        """Bulk construction of a recall-competent HNSW graph (SURVEY 8f rank 1 + 3): heuristic forward links over the
        ef_construction exact nearest + reverse links, cut to 2 M; upper layers; entry = first node of the highest level.
        Search it with search_graph(ef > 0, entry = ENTRY_DESCEND)."""
        check(self._L.zvdb_hip_build_graph(self._h, M, ef_construction, mode, level_seed))

    def graph_info(self) -> Tuple[int, int]:
        e, ml = C.c_uint64(0), C.c_uint32(0)
        check(self._L.zvdb_hip_graph_info(self._h, C.byref(e), C.byref(ml)))
        return e.value, ml.value

    def save(self, path: str) -> None:
        """Write the index (rows as stored, every layer of the graph) to one flat file (include/zvdb_hip.h: on-disk format)."""
        check(self._L.zvdb_hip_save(self._h, os.fsencode(path)))

    @classmethod
    def load(cls, path: str, device: int = -1) -> "Index":
        L = _lib.hip()
        h = C.c_void_p()
        check(L.zvdb_hip_load(os.fsencode(path), device, C.byref(h)))
        self = cls._from_handle(h.value)
        self._own = True
        return self

    def storage_info(self) -> Tuple[int, int]:
        """(bytes of one stored row, bytes resident on the device for this index)."""
        rb, tb = C.c_uint64(0), C.c_uint64(0)
        check(self._L.zvdb_hip_storage_info(self._h, C.byref(rb), C.byref(tb)))
        return rb.value, tb.value

    def get_neighbors_layer(self, node: int, layer: int) -> np.ndarray:
        out = np.zeros(4096, dtype=np.uint64)
        n = C.c_uint32(0)
        check(self._L.zvdb_hip_get_neighbors_layer(self._h, node, layer, out.ctypes.data_as(_u64p), out.size, C.byref(n)))
        return out[:n.value].copy()

    def get_neighbors(self, node: int) -> np.ndarray:
        out = np.zeros(4096, dtype=np.uint64)
        n = C.c_uint32(0)
        check(self._L.zvdb_hip_get_neighbors(self._h, node, out.ctypes.data_as(_u64p), out.size, C.byref(n)))
        return out[:n.value].copy()

    # ---- a1/a2/a3 seam -------------------------------------------------------------
    def distances(self, query, ids) -> np.ndarray:
        q = _f32(query).reshape(-1)
        a = np.ascontiguousarray(ids, dtype=np.uint64)
        out = np.zeros(a.size, dtype=np.float32)
        check(self._L.zvdb_hip_distances(self._h, q.ctypes.data_as(_f32p), q.size, a.ctypes.data_as(_u64p),
                                         a.size, out.ctypes.data_as(_f32p)))
        return out

    def distances_from(self, owner: int, ids) -> np.ndarray:
        a = np.ascontiguousarray(ids, dtype=np.uint64)
        out = np.zeros(a.size, dtype=np.float32)
        check(self._L.zvdb_hip_distances_from(self._h, owner, a.ctypes.data_as(_u64p), a.size,
                                              out.ctypes.data_as(_f32p)))
        return out

    def greedy_layer0(self, owner: int, ep: int, ep_dist: float) -> Tuple[int, float, int]:
        e, d, ev = C.c_uint64(ep), C.c_float(ep_dist), C.c_uint64(0)
        check(self._L.zvdb_hip_greedy_layer0(self._h, owner, C.byref(e), C.byref(d), C.byref(ev)))
        return e.value, d.value, ev.value

    # ---- search ----------------------------------------------------------------------
    def search_graph(self, queries, k: int, ef: int = 0, entry: int = 0, with_stats: bool = False):
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq, dim = q.shape
        ids = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        dist = np.zeros((nq, max(k, 1)), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        st = SearchStats()
        check(self._L.zvdb_hip_search_graph(self._h, q.ctypes.data_as(_f32p), nq, dim, k, ef, entry,
                                            ids.ctypes.data_as(_u64p), dist.ctypes.data_as(_f32p),
                                            cnt.ctypes.data_as(_u32p), C.byref(st)))
        if with_stats:
            return ids[:, :k], dist[:, :k], cnt, (st.dist_evals, st.pops)
        return ids[:, :k], dist[:, :k], cnt

    def search_bruteforce(self, queries, k: int, mode: int = _lib.BF_F32):
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq, dim = q.shape
        ids = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        dist = np.zeros((nq, max(k, 1)), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        check(self._L.zvdb_hip_search_bruteforce(self._h, q.ctypes.data_as(_f32p), nq, dim, k, mode,
                                                 ids.ctypes.data_as(_u64p), dist.ctypes.data_as(_f32p),
                                                 cnt.ctypes.data_as(_u32p)))
        return ids[:, :k], dist[:, :k], cnt

    # ---- device-pointer variants -------------------------------------------------------
    def bruteforce_workspace(self, nq: int, k: int, mode: int) -> int:
        b = C.c_uint64(0)
        check(self._L.zvdb_hip_bruteforce_workspace(self._h, nq, k, mode, C.byref(b)))
        return b.value

    def search_bruteforce_dev(self, q_ptr: int, nq: int, dim: int, k: int, mode: int, id_offset: int,
                              ws_ptr: int, ws_bytes: int, ids_ptr: int, dist_ptr: int, cnt_ptr: int,
                              stream: int = 0) -> None:
        check(self._L.zvdb_hip_search_bruteforce_dev(self._h, q_ptr, nq, dim, k, mode, id_offset, ws_ptr,
                                                     ws_bytes, ids_ptr, dist_ptr, cnt_ptr, stream))

    def search_graph_dev(self, q_ptr: int, nq: int, dim: int, k: int, ef: int, entry: int, id_offset: int,
                         ids_ptr: int, dist_ptr: int, cnt_ptr: int, stats_ptr: int = 0, stream: int = 0) -> None:
        check(self._L.zvdb_hip_search_graph_dev(self._h, q_ptr, nq, dim, k, ef, entry, id_offset, ids_ptr,
                                                dist_ptr, cnt_ptr, stats_ptr, stream))

    def search_graph_reserve(self, nq: int, k: int, ef: int) -> None:
        """Takes every index-owned buffer a later search_graph_dev of this shape would allocate on first need."""
        check(self._L.zvdb_hip_search_graph_reserve(self._h, nq, k, ef))

    def set_shard_count(self, shards: int) -> None:
        """This index is one of `shards` id-range shards (zvdb_hip_index_set_shard_count): the fp4 screen keeps its
        share of the survivor budget."""
        check(self._L.zvdb_hip_index_set_shard_count(self._h, int(shards)))

    def set_option(self, option: int, value: int) -> None:
        """zvdb_hip_index_set_option (OPT_CERTIFY, OPT_CERT_MIN_LEVEL, ...)."""
        check(self._L.zvdb_hip_index_set_option(self._h, int(option), int(value)))

    def cert_info(self) -> dict:
        """Certificate counters of the exact brute-force screens: queries, batches, repaired (exact-scan recomputations), level."""
        from ._lib import CertInfo
        ci = CertInfo()
        check(self._L.zvdb_hip_bruteforce_cert_info(self._h, C.byref(ci)))
        return {"queries": ci.queries, "batches": ci.batches, "repaired": ci.repaired, "level": ci.level,
                "enabled": bool(ci.enabled), "rerank_rows": ci.rerank_rows, "rerank_cycles": list(ci.rerank_cycles),
                "guess_misses": ci.guess_misses, "bound_checked": ci.bound_checked, "bound_violations": ci.bound_violations}

    def set_screen_survivors(self, survivors: int) -> None:
        """Minimum number of fp4-screen survivors re-ranked exactly per query (0 = from k; 64, 128 or 256)."""
        check(self._L.zvdb_hip_index_set_screen_survivors(self._h, int(survivors)))

    def timing_enable(self, on=True) -> None:
        """True / 1: start a new measurement; 2: resume (sampled timing); False / 0: pause (samples stay readable)."""
        check(self._L.zvdb_hip_timing_enable(self._h, int(on)))

    def timing_read(self) -> Tuple[float, int]:
        ms, n = C.c_double(0), C.c_uint32(0)
        check(self._L.zvdb_hip_timing_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def fill_synthetic_dev(ptr: int, nrows: int, dim: int, seed: int, first_row: int = 0, stream: int = 0) -> None:
    check(_lib.hip().zvdb_hip_fill_synthetic_dev(ptr, nrows, dim, seed, first_row, stream))


def merge_topk_dev(ids_ptr: int, dist_ptr: int, cnt_ptr: int, G: int, nq: int, k: int, out_ids: int,
                   out_dist: int, out_cnt: int, stream: int = 0) -> None:
    check(_lib.hip().zvdb_hip_merge_topk_dev(ids_ptr, dist_ptr, cnt_ptr, G, nq, k, out_ids, out_dist, out_cnt,
                                             stream))


def packed_result_bytes(nq: int, k: int) -> int:
    """Bytes of one rank's packed result chunk (ids u64 | dist f32 | count u32), padded to 16."""
    return (nq * k * 12 + nq * 4 + 15) // 16 * 16


def merge_topk_packed_dev(packed_ptr: int, rank_stride: int, G: int, nq: int, k: int, out_ids: int, out_dist: int,
                          out_cnt: int, stream: int = 0) -> None:
    check(_lib.hip().zvdb_hip_merge_topk_packed_dev(packed_ptr, rank_stride, G, nq, k, out_ids, out_dist, out_cnt,
                                                    stream))
