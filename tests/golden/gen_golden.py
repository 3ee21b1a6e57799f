"""Generates tests/golden/golden_small.npz: small seeded inputs with the outputs the ORACLE (oracle/zvdb_oracle.c, the CPU
restatement of src/hnsw.zig) gives for them, plus the known answers of the reference's own tests.  The reference (Zig) cannot
be built in this image (no zig toolchain), so there are no reference-generated vectors: the fixture pins the
oracle AND the HIP path against silent drift, and the reference's hand-checkable answers (src/test_hnsw.zig:24-41) anchor it.

    python tests/golden/gen_golden.py        # rewrites golden_small.npz (deterministic)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

oracle.build()
out = {}
# (1) the reference's Basic test: three points, query [3,4,5], k = 2 -> ids [1, 0], squared distances [3, 12]
out["basic_points"] = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], dtype=np.float32)
out["basic_query"] = np.array([3, 4, 5], dtype=np.float32)
out["basic_ids"] = np.array([1, 0], dtype=np.uint64)
out["basic_dist"] = np.array([3.0, 12.0], dtype=np.float32)
# (2) a seeded 400 x 24 index, m = 4: graph on every layer, searches, counters
n, d, m = 400, 24, 4
V = oracle.uniform_rows(0x5EED, 0, n, d)
Q = oracle.uniform_rows(0xC0FFEE, 0, 12, d)
h = oracle.HNSW(m, 200, level_seed=7)
for r in V:
    h.insert(r)
levels = np.array([h.node_level(i) for i in range(n)], dtype=np.int32)
adj = np.full((n, int(levels.max()) + 1, m), -1, dtype=np.int64)
for i in range(n):
    for layer in range(levels[i] + 1):
        nb = h.neighbors(i, layer)
        adj[i, layer, :len(nb)] = nb
out.update(g_rows_seed=np.uint64(0x5EED), g_query_seed=np.uint64(0xC0FFEE), g_n=np.int64(n), g_d=np.int64(d), g_m=np.int64(m),
           g_levels=levels, g_adj=adj)
for k in (1, 10, 25):
    ids = np.full((len(Q), k), -1, dtype=np.int64)
    dist = np.full((len(Q), k), np.inf, dtype=np.float32)
    cnt = np.zeros(len(Q), dtype=np.int32)
    h.reset_counters()
    for qi, q in enumerate(Q):
        ri, rd = h.search(q, k)
        ids[qi, :len(ri)] = ri
        dist[qi, :len(ri)] = rd
        cnt[qi] = len(ri)
    c = h.counters()
    out[f"g_search_ids_k{k}"] = ids
    out[f"g_search_dist_k{k}"] = dist
    out[f"g_search_cnt_k{k}"] = cnt
    out[f"g_search_evals_pops_k{k}"] = np.array([c.search_dist_evals, c.search_pops], dtype=np.int64)
# (3) exact brute-force top-10 over the same rows (L2 and dot)
for name, metric in (("l2", 0), ("dot", 1)):
    ri, rd, rc = oracle.bruteforce_topk(V, Q, 10, metric=metric)
    out[f"bf_{name}_ids"] = ri
    out[f"bf_{name}_dist"] = rd
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_small.npz"), **out)
print("wrote golden_small.npz:", {k: v.shape for k, v in out.items()})
