"""CPU ORACLE -- test infrastructure, NOT product code.

ctypes front-end of oracle/zvdb_oracle.c (a C restatement of the reference's
src/hnsw.zig hot path).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; nothing under zvdb_amd/ does.

Parity-pin status: pinned by the known answers of the reference's own tests
(src/test_hnsw.zig, see tests/test_oracle_pins.py).  Exact-tie pop order and
every extension without a reference symbol (brute force, ef mode, dot/cosine,
bf16) are "parity unpinned" -- see oracle/zvdb_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libzvdb_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (strict fp32).  Building the checker is not using it."""
    src = os.path.join(_HERE, "zvdb_oracle.c")
    hdr = os.path.join(_HERE, "zvdb_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(
            ["gcc", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c11",
             "-shared", "-o", _LIB_PATH, src])
    return _LIB_PATH


class Counters(C.Structure):
    _fields_ = [("search_dist_evals", C.c_uint64), ("search_pops", C.c_uint64),
                ("sort_dist_evals", C.c_uint64), ("insert_dist_evals", C.c_uint64),
                ("shrink_dist_evals", C.c_uint64)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    f32p, u64p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    L.orc_distance_f32.restype = C.c_float
    L.orc_distance_f32.argtypes = [f32p, f32p, C.c_size_t]
    L.orc_distance_f64.restype = C.c_double
    L.orc_distance_f64.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t]
    L.orc_distance_i32.restype = C.c_int32
    L.orc_distance_i32.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_size_t]
    L.orc_distance_blocked16_f32.restype = C.c_float
    L.orc_distance_blocked16_f32.argtypes = [f32p, f32p, C.c_size_t]
    L.orc_mix64.restype = C.c_uint64
    L.orc_mix64.argtypes = [C.c_uint64, C.c_uint64]
    L.orc_fill_uniform.restype = None
    L.orc_fill_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, f32p]
    L.orc_bf16_round.restype = C.c_float
    L.orc_bf16_round.argtypes = [C.c_float]
    L.orc_hnsw_create.restype = C.c_void_p
    L.orc_hnsw_create.argtypes = [C.c_size_t, C.c_size_t, C.c_uint64]
    L.orc_hnsw_destroy.restype = None
    L.orc_hnsw_destroy.argtypes = [C.c_void_p]
    L.orc_hnsw_insert.restype = C.c_int
    L.orc_hnsw_insert.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_int]
    L.orc_hnsw_search.restype = C.c_size_t
    L.orc_hnsw_search.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_size_t, u64p, f32p]
    L.orc_hnsw_search_many.restype = None
    L.orc_hnsw_search_many.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_size_t, C.c_size_t, u64p, f32p, u64p]
    L.orc_hnsw_search_ef_ex.restype = C.c_size_t
    L.orc_hnsw_search_ef_ex.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t, u64p, f32p]
    L.orc_hnsw_search_ef_w.restype = C.c_size_t
    L.orc_hnsw_search_ef_w.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, u64p, f32p]
    L.orc_hnsw_search_ef.restype = C.c_size_t
    L.orc_hnsw_search_ef.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_size_t, C.c_size_t, u64p, f32p]
    L.orc_hnsw_count.restype = C.c_size_t
    L.orc_hnsw_count.argtypes = [C.c_void_p]
    L.orc_hnsw_dim.restype = C.c_size_t
    L.orc_hnsw_dim.argtypes = [C.c_void_p]
    L.orc_hnsw_max_level.restype = C.c_size_t
    L.orc_hnsw_max_level.argtypes = [C.c_void_p]
    L.orc_hnsw_node_level.restype = C.c_int
    L.orc_hnsw_node_level.argtypes = [C.c_void_p, C.c_size_t]
    L.orc_hnsw_neighbors.restype = C.c_size_t
    L.orc_hnsw_neighbors.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, u64p, C.c_size_t]
    L.orc_hnsw_point.restype = f32p
    L.orc_hnsw_point.argtypes = [C.c_void_p, C.c_size_t]
    L.orc_hnsw_set_neighbors.restype = C.c_int
    L.orc_hnsw_set_neighbors.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, u64p, C.c_size_t]
    L.orc_hnsw_counters.restype = None
    L.orc_hnsw_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
    L.orc_hnsw_reset_counters.restype = None
    L.orc_hnsw_reset_counters.argtypes = [C.c_void_p]
    L.orc_bruteforce_topk.restype = None
    L.orc_bruteforce_topk.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_size_t, f32p, C.c_size_t,
                                      C.c_size_t, C.c_int, u64p, f32p, u32p]
    L.orc_bulk_level.restype = C.c_uint32
    L.orc_bulk_level.argtypes = [C.c_uint64, C.c_uint64]
    L.orc_graph_build_layer0.restype = None
    L.orc_graph_build_layer0.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, u64p, f32p, C.c_size_t, C.c_size_t,
                                         C.c_size_t, C.c_size_t, u32p, u32p]
    L.orc_hnsw_descend.restype = C.c_size_t
    L.orc_hnsw_descend.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_int, C.c_size_t]
    L.orc_merge_topk.restype = None
    L.orc_merge_topk.argtypes = [u64p, f32p, u32p, C.c_size_t, C.c_size_t, C.c_size_t, u64p, f32p, u32p]
    _lib = L
    return L


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def distance(a, b) -> np.float32:
    a, b = _f32(a), _f32(b)
    assert a.shape == b.shape and a.ndim == 1
    return np.float32(lib().orc_distance_f32(_p(a, C.c_float), _p(b, C.c_float), a.size))


def distance_blocked16(a, b) -> np.float32:
    a, b = _f32(a), _f32(b)
    return np.float32(lib().orc_distance_blocked16_f32(_p(a, C.c_float), _p(b, C.c_float), a.size))


def distance_f64(a, b) -> float:
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return float(lib().orc_distance_f64(_p(a, C.c_double), _p(b, C.c_double), a.size))


def distance_i32(a, b) -> int:
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    return int(lib().orc_distance_i32(_p(a, C.c_int32), _p(b, C.c_int32), a.size))


def fill_uniform(seed: int, first_elem: int, count: int) -> np.ndarray:
    out = np.empty(count, dtype=np.float32)
    lib().orc_fill_uniform(seed, first_elem, count, _p(out, C.c_float))
    return out


def uniform_rows(seed: int, first_row: int, nrows: int, dim: int) -> np.ndarray:
    """Rows [first_row, first_row+nrows) of the synthetic U[0,1) corpus `seed`."""
    return fill_uniform(seed, first_row * dim, nrows * dim).reshape(nrows, dim)


def bf16_round(a) -> np.ndarray:
    a = _f32(a)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(a.shape)


class HNSW:
    """Oracle mirror of HNSW(f32) (src/hnsw.zig:8-247)."""

    def __init__(self, m: int = 16, ef_construction: int = 200, level_seed: int = 1):
        self._L = lib()
        self._h = self._L.orc_hnsw_create(m, ef_construction, level_seed)
        if not self._h:
            raise MemoryError
        self.m = m
        self.ef_construction = ef_construction

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.orc_hnsw_destroy(h)

    def insert(self, point, level: int = -1) -> None:
        p = _f32(point)
        rc = self._L.orc_hnsw_insert(self._h, _p(p, C.c_float), p.size, level)
        if rc == -2:
            raise ValueError("Mismatched dimensions in distance calculation")
        if rc:
            raise RuntimeError(f"oracle insert failed rc={rc}")

    def search(self, query, k: int) -> Tuple[np.ndarray, np.ndarray]:
        q = _f32(query)
        ids = np.zeros(max(k, 1), dtype=np.uint64)
        dist = np.zeros(max(k, 1), dtype=np.float32)
        n = self._L.orc_hnsw_search(self._h, _p(q, C.c_float), q.size, k,
                                    _p(ids, C.c_uint64), _p(dist, C.c_float))
        return ids[:n].copy(), dist[:n].copy()

    def search_many(self, queries, k: int, threads: int = 1):
        """k-NN of every row of `queries` (reference-compat search, no counters), the queries split over `threads` OS threads
        (ctypes releases the GIL; the search is read-only).  Returns ids [nq, k], dist [nq, k], n [nq]."""
        import threading
        Q = np.ascontiguousarray(queries, dtype=np.float32)
        nq, d = Q.shape
        ids = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        dist = np.zeros((nq, max(k, 1)), dtype=np.float32)
        n = np.zeros(nq, dtype=np.uint64)
        threads = max(1, min(threads, nq))
        parts = [(t * nq // threads, (t + 1) * nq // threads) for t in range(threads)]

        def work(a, b):
            if b > a:
                self._L.orc_hnsw_search_many(self._h, _p(Q[a:b], C.c_float), b - a, d, k, _p(ids[a:b], C.c_uint64),
                                             _p(dist[a:b], C.c_float), _p(n[a:b], C.c_uint64))
        th = [threading.Thread(target=work, args=p) for p in parts]
        for x in th:
            x.start()
        for x in th:
            x.join()
        return ids, dist, n

    def search_ef(self, query, k: int, ef: int, metric: int = 0, entry: int = 0, width: int = 1) -> Tuple[np.ndarray, np.ndarray]:
        """ef-mode beam search (extension; the spec of k_search_ef): metric 0 = squared L2, 1 = inner product; width = nodes
        expanded per step (orc_hnsw_search_ef_w; 1 = the plain search)."""
        q = _f32(query)
        ids = np.zeros(max(k, 1), dtype=np.uint64)
        dist = np.zeros(max(k, 1), dtype=np.float32)
        n = self._L.orc_hnsw_search_ef_w(self._h, _p(q, C.c_float), q.size, k, ef, metric, entry, width,
                                         _p(ids, C.c_uint64), _p(dist, C.c_float))
        return ids[:n].copy(), dist[:n].copy()

    def descend(self, query, metric: int = 0, top: int = 0) -> int:
        """Hierarchical entry descent of the bulk graph (extension; the spec of k_descend): layer-0 entry of the beam."""
        q = _f32(query)
        return int(self._L.orc_hnsw_descend(self._h, _p(q, C.c_float), q.size, metric, top))

    def count(self) -> int:
        return int(self._L.orc_hnsw_count(self._h))

    @property
    def dim(self) -> int:
        return int(self._L.orc_hnsw_dim(self._h))

    @property
    def max_level(self) -> int:
        return int(self._L.orc_hnsw_max_level(self._h))

    def node_level(self, i: int) -> int:
        return int(self._L.orc_hnsw_node_level(self._h, i))

    def neighbors(self, i: int, layer: int = 0) -> Optional[np.ndarray]:
        cap = 4096
        out = np.zeros(cap, dtype=np.uint64)
        n = self._L.orc_hnsw_neighbors(self._h, i, layer, _p(out, C.c_uint64), cap)
        if n == C.c_size_t(-1).value:
            return None
        return out[:n].copy()

    def set_neighbors(self, i: int, layer: int, ids) -> None:
        a = np.ascontiguousarray(ids, dtype=np.uint64)
        rc = self._L.orc_hnsw_set_neighbors(self._h, i, layer, _p(a, C.c_uint64), a.size)
        if rc:
            raise KeyError("NodeNotFound")

    def point(self, i: int) -> np.ndarray:
        p = self._L.orc_hnsw_point(self._h, i)
        return np.ctypeslib.as_array(p, shape=(self.dim,)).copy()

    def layer0(self, stride: int) -> Tuple[np.ndarray, np.ndarray]:
        """Fixed-stride layer-0 adjacency (adj[N][stride] u32, deg[N] u32)."""
        n = self.count()
        adj = np.zeros((n, stride), dtype=np.uint32)
        deg = np.zeros(n, dtype=np.uint32)
        for i in range(n):
            nb = self.neighbors(i, 0)
            deg[i] = len(nb)
            adj[i, :len(nb)] = nb
        return adj, deg

    def counters(self) -> Counters:
        c = Counters()
        self._L.orc_hnsw_counters(self._h, C.byref(c))
        return c

    def reset_counters(self) -> None:
        self._L.orc_hnsw_reset_counters(self._h)


def bruteforce_topk(V, Q, k: int, metric: int = 0):
    V, Q = _f32(V), _f32(Q)
    N, d = V.shape
    nq = Q.shape[0]
    ids = np.zeros((nq, k), dtype=np.uint64)
    dist = np.zeros((nq, k), dtype=np.float32)
    cnt = np.zeros(nq, dtype=np.uint32)
    lib().orc_bruteforce_topk(_p(V, C.c_float), N, d, d, _p(Q, C.c_float), nq, k, metric,
                              _p(ids, C.c_uint64), _p(dist, C.c_float), _p(cnt, C.c_uint32))
    return ids, dist, cnt


def bruteforce_topk_synthetic(seed: int, n: int, dim: int, Q, k: int, metric: int = 0, slice_rows: int = 125_000, threads: int = 16):
    """Exact top-k of every query of Q over ALL n rows of the synthetic corpus `seed`, without holding the corpus: the rows are
    regenerated slice by slice (uniform_rows), each slice answered by bruteforce_topk (hnsw.zig:182-192 distances, ties by lower
    id), the per-slice lists merged by merge_topk.  Slices run on `threads` host threads (the C calls release the GIL).
    10M x 768 x 16 queries: ~30 s on 16 cores.  Test / bench checker only."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    Q = _f32(Q)
    nq = Q.shape[0]
    firsts = list(range(0, n, slice_rows))
    threads = max(1, min(threads, len(firsts), os.cpu_count() or 1))

    def work(first):
        rows = min(slice_rows, n - first)
        V = uniform_rows(seed, first, rows, dim)
        i, d, c = bruteforce_topk(V, Q, k, metric)
        i += np.uint64(first)
        return i, d, c

    with ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(work, firsts))
    ids = np.stack([p[0] for p in parts])
    dist = np.stack([p[1] for p in parts])
    cnt = np.stack([p[2] for p in parts])
    return merge_topk(ids, dist, cnt, k)


def merge_topk(ids, dist, count, k: int):
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    dist = _f32(dist)
    count = np.ascontiguousarray(count, dtype=np.uint32)
    G, nq, kk = ids.shape
    assert kk == k
    oi = np.zeros((nq, k), dtype=np.uint64)
    od = np.zeros((nq, k), dtype=np.float32)
    oc = np.zeros(nq, dtype=np.uint32)
    lib().orc_merge_topk(_p(ids, C.c_uint64), _p(dist, C.c_float), _p(count, C.c_uint32), G, nq, k,
                         _p(oi, C.c_uint64), _p(od, C.c_float), _p(oc, C.c_uint32))
    return oi, od, oc


# ---- bulk graph construction (extension; the specification of zvdb_hip_build_graph: oracle/zvdb_oracle.c) ------------------
def bulk_levels(seed: int, n: int) -> np.ndarray:
    L = lib()
    return np.array([L.orc_bulk_level(seed, i) for i in range(n)], dtype=np.uint32)


def _knn_excluding_self(V, rows, k: int, metric: int, threads: int = 8):
    """For every row of V[rows] its k nearest among V[rows] by (distance, id), itself excluded: the row's own entry is
    removed from the top-(k+1); if it is not among them (more than k exact duplicates ahead of it) the last entry goes."""
    import threading
    sub = np.ascontiguousarray(V[rows])
    n = len(rows)
    kk = min(k + 1, n)
    ids = np.zeros((n, kk), dtype=np.uint64)
    dist = np.zeros((n, kk), dtype=np.float32)
    parts = [(t * n // threads, (t + 1) * n // threads) for t in range(threads)]

    def work(a, b):
        if b > a:
            i, d, _ = bruteforce_topk(sub, sub[a:b], kk, metric)
            ids[a:b], dist[a:b] = i, d

    th = [threading.Thread(target=work, args=p) for p in parts]
    [t.start() for t in th]
    [t.join() for t in th]
    kout = min(k, n - 1)
    oi = np.zeros((n, max(kout, 1)), dtype=np.uint64)
    od = np.zeros((n, max(kout, 1)), dtype=np.float32)
    for r in range(n):
        keep = [j for j in range(kk) if ids[r, j] != r][:kout]
        oi[r, :kout] = ids[r, keep]
        od[r, :kout] = dist[r, keep]
    return oi[:, :kout], od[:, :kout]


def graph_build_layer0(V, cand_ids, cand_dist, M: int, M0: int, cmax: int, metric: int = 0):
    V = _f32(V)
    N, d = V.shape
    cand_ids = np.ascontiguousarray(cand_ids, dtype=np.uint64)
    cand_dist = _f32(cand_dist)
    adj = np.zeros((N, M0), dtype=np.uint32)
    deg = np.zeros(N, dtype=np.uint32)
    lib().orc_graph_build_layer0(_p(V, C.c_float), N, d, d, metric, _p(cand_ids, C.c_uint64), _p(cand_dist, C.c_float),
                                 cand_ids.shape[1], M, M0, cmax, _p(adj, C.c_uint32), _p(deg, C.c_uint32))
    return adj, deg


def build_bulk_graph(V, M: int, ef_construction: int, level_seed: int, metric: int = 0, cmax: int = 128):
    """The whole bulk graph as the specification defines it: levels from the counter hash, layer 0 = heuristic selection over
    the ef_construction exact nearest + reverse links, capped at M0 = 2 M; every upper layer L = the M exact nearest among
    the nodes whose level reaches L.  Returns (HNSW holding points, levels and every list; top entry node)."""
    V = _f32(V)
    N = V.shape[0]
    levels = bulk_levels(level_seed, N)
    h = HNSW(2 * M, ef_construction)
    for i in range(N):
        h.insert(V[i], int(levels[i]))
    ci, cd = _knn_excluding_self(V, np.arange(N), min(ef_construction, N - 1), metric)
    adj, deg = graph_build_layer0(V, ci, cd, M, 2 * M, cmax, metric)
    for i in range(N):
        h.set_neighbors(i, 0, adj[i, :deg[i]])
    for L in range(1, int(levels.max()) + 1):
        rows = np.nonzero(levels >= L)[0]
        if len(rows) > 1:
            ui, _ = _knn_excluding_self(V, rows, M, metric)
            for r, node in enumerate(rows):
                h.set_neighbors(int(node), L, rows[ui[r].astype(np.int64)])
        else:
            for node in rows:
                h.set_neighbors(int(node), L, [])
    top = int(np.argmax(levels))  # the first node of the highest level
    return h, levels, top
