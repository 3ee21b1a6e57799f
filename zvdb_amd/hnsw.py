"""`HNSW` -- Python face of the C++ host mirror (host/hnsw.hpp) of zvdb's `HNSW(T)`
(reference src/hnsw.zig:8-247, re-exported at src/zvdb.zig:1).

Same surface as the reference: ``HNSW(m, ef_construction)`` (no dimension at init, hnsw.zig:52),
``insert(point)`` (hnsw.zig:73), ``search(query, k) -> [Node]`` (hnsw.zig:194) whose length may be
< k (test_hnsw.zig:122-125), ``count()`` (= ``hnsw.nodes.count()``, test_hnsw.zig:198).  The graph
bookkeeping is the C++ host code; every distance / scan / top-k is a HIP kernel.  No CPU fallback."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import _lib


@dataclass
class Node:
    """By-value copy of a reference Node (hnsw.zig:12-17): id, point, connections."""
    id: int
    point: np.ndarray
    connections: List[List[int]]
    distance: float  # squared L2 to the query (what the reference's tests recompute)


class HNSW:
    def __init__(self, m: int = 16, ef_construction: int = 200, device: int = -1, flags: int = 0):
        self._L = _lib.host()
        self._p = self._L.zvdb_host_hnsw_init(m, ef_construction, device, flags)
        if not self._p:
            raise MemoryError("OutOfMemory")
        self.m = m
        self.ef_construction = ef_construction
        self.last_evals = 0
        self.last_pops = 0

    # deinit (hnsw.zig:64-71)
    def deinit(self) -> None:
        p, self._p = self._p, None
        if p:
            self._L.zvdb_host_hnsw_deinit(p)

    def __del__(self):
        try:
            self.deinit()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc == 0:
            return
        msg = self._L.zvdb_host_last_error().decode(errors="replace")
        if rc == 100:
            raise _lib.ZvdbPanic(msg)
        raise _lib.ZvdbError(rc, msg)

    def insert(self, point, _level: int = -1) -> None:
        p = np.ascontiguousarray(point, dtype=np.float32).reshape(-1)
        self._check(self._L.zvdb_host_hnsw_insert(self._p, p.ctypes.data_as(C.POINTER(C.c_float)), p.size,
                                                  _level))

    def insert_many(self, points, levels=None) -> None:
        """n inserts with the result of n insert() calls in this order; the layer-0 work of a chunk is one device launch
        (zvdb_hip_insert_batch_layer0).  Extension: the reference inserts one point per call."""
        pts = np.ascontiguousarray(points, dtype=np.float32)
        if pts.ndim != 2:
            raise ValueError("points must be [n][dim]")
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.int32)
        self._check(self._L.zvdb_host_hnsw_insert_many(
            self._p, pts.ctypes.data_as(C.POINTER(C.c_float)), pts.shape[0], pts.shape[1],
            None if lv is None else lv.ctypes.data_as(C.POINTER(C.c_int))))

    def search(self, query, k: int) -> List[Node]:
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        ids = np.zeros(max(k, 1), dtype=np.uint64)
        dist = np.zeros(max(k, 1), dtype=np.float32)
        n = C.c_size_t(0)
        ev, pp = C.c_uint64(0), C.c_uint64(0)
        self._check(self._L.zvdb_host_hnsw_search(
            self._p, q.ctypes.data_as(C.POINTER(C.c_float)), q.size, k,
            ids.ctypes.data_as(C.POINTER(C.c_uint64)), dist.ctypes.data_as(C.POINTER(C.c_float)),
            C.byref(n), C.byref(ev), C.byref(pp)))
        self.last_evals, self.last_pops = ev.value, pp.value
        return [Node(int(ids[i]), self.point(int(ids[i])), self.connections_all(int(ids[i])), float(dist[i]))
                for i in range(n.value)]

    def search_ids(self, query, k: int):
        """ids/distances only (no Node copies): what parity tests compare."""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        ids = np.zeros(max(k, 1), dtype=np.uint64)
        dist = np.zeros(max(k, 1), dtype=np.float32)
        n = C.c_size_t(0)
        ev, pp = C.c_uint64(0), C.c_uint64(0)
        self._check(self._L.zvdb_host_hnsw_search(
            self._p, q.ctypes.data_as(C.POINTER(C.c_float)), q.size, k,
            ids.ctypes.data_as(C.POINTER(C.c_uint64)), dist.ctypes.data_as(C.POINTER(C.c_float)),
            C.byref(n), C.byref(ev), C.byref(pp)))
        self.last_evals, self.last_pops = ev.value, pp.value
        return ids[:n.value].copy(), dist[:n.value].copy()

    def search_many(self, queries, k: int):
        """nq searches in ONE device launch: (ids [nq][k] u64, dist [nq][k] f32, count [nq] u32), the results of nq search() calls
        without the Node copies (HNSW::search_many, hnsw.hpp; `searchMany` of bindings/hnsw_hip.zig).  Extension: hnsw.zig:194 takes
        one query per call."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2:
            raise ValueError("queries must be [nq][dim]")
        nq = q.shape[0]
        ids = np.full((nq, max(k, 1)), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
        dist = np.full((nq, max(k, 1)), np.inf, dtype=np.float32)
        cnt = np.zeros(max(nq, 1), dtype=np.uint32)
        self._check(self._L.zvdb_host_hnsw_search_many(
            self._p, q.ctypes.data_as(C.POINTER(C.c_float)), nq, q.shape[1], k,
            ids.ctypes.data_as(C.POINTER(C.c_uint64)), dist.ctypes.data_as(C.POINTER(C.c_float)), cnt.ctypes.data_as(C.POINTER(C.c_uint32))))
        return ids[:, :k], dist[:, :k], cnt[:nq]

    def count(self) -> int:
        return int(self._L.zvdb_host_hnsw_count(self._p))

    @property
    def dim(self) -> int:
        return int(self._L.zvdb_host_hnsw_dim(self._p))

    @property
    def max_level(self) -> int:
        return int(self._L.zvdb_host_hnsw_max_level(self._p))

    @property
    def entry_point(self) -> Optional[int]:
        e = int(self._L.zvdb_host_hnsw_entry_point(self._p))
        return None if e < 0 else e

    def connections(self, node: int, layer: int = 0) -> Optional[np.ndarray]:
        cap = 8192
        out = np.zeros(cap, dtype=np.uint64)
        n = int(self._L.zvdb_host_hnsw_connections(self._p, node, layer, out.ctypes.data_as(C.POINTER(C.c_uint64)), cap))
        if n == -2:
            raise KeyError("NodeNotFound")
        return None if n < 0 else out[:n].copy()

    def connections_all(self, node: int) -> List[List[int]]:
        out, layer = [], 0
        while True:
            c = self.connections(node, layer)
            if c is None:
                return out
            out.append([int(x) for x in c])
            layer += 1

    def point(self, node: int) -> np.ndarray:
        out = np.zeros(self.dim, dtype=np.float32)
        h = self._L.zvdb_host_hnsw_handle(self._p)
        _lib.check(_lib.hip().zvdb_hip_get_vector(h, node, out.ctypes.data_as(C.POINTER(C.c_float)), self.dim))
        return out

    @property
    def index(self):
        """The underlying device index as a low-level `Index` view (not owning)."""
        from .index import Index
        return Index._from_handle(self._L.zvdb_host_hnsw_handle(self._p))
