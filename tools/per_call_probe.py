# the reference's calling pattern (shared_benchmarks.zig:68-72, :104-109: one point / one query per call) through the drop-in API:
# wall time per call against the kernels' own time (run under rocprofv3 --kernel-trace --stats: tools/prof_per_call.sh)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, zvdb_amd as zv
P, D, NQ, K = 100_000, 128, 3000, 10
rng = np.random.default_rng(1)
V = rng.random((P, D), dtype=np.float32); Q = rng.random((NQ, D), dtype=np.float32)
h = zv.HNSW(16, 200)
h.insert_many(V[:P - 5000], np.zeros(P - 5000, dtype=np.int32))
t0 = time.perf_counter()
for r in V[P - 5000:]: h.insert(r, _level=0)
ti = (time.perf_counter() - t0) / 5000
for q in Q[:200]: h.search_ids(q, K)
t0 = time.perf_counter()
for q in Q: h.search_ids(q, K)
ts = (time.perf_counter() - t0) / NQ
print(f"per call through the Python mirror: insert {ti * 1e6:.1f} us, search {ts * 1e6:.1f} us (100k x 128, k = 10)")
import ctypes as C
from zvdb_amd import _lib
L = _lib.hip(); idx = h.index
ids = np.zeros(K, dtype=np.uint64); dist = np.zeros(K, dtype=np.float32); cnt = np.zeros(1, dtype=np.uint32)
args = (idx._h, None, 1, D, K, 0, 0, ids.ctypes.data_as(C.POINTER(C.c_uint64)), dist.ctypes.data_as(C.POINTER(C.c_float)), cnt.ctypes.data_as(C.POINTER(C.c_uint32)), None)
fn = L.zvdb_hip_search_graph
qp = [q.ctypes.data_as(C.POINTER(C.c_float)) for q in Q]
t0 = time.perf_counter()
for p in qp: fn(idx._h, p, 1, D, K, 0, 0, args[7], args[8], args[9], None)
print(f"  search straight through the C ABI (ctypes, preallocated outputs): {(time.perf_counter() - t0) / NQ * 1e6:.1f} us")
