import time, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle
V = oracle.uniform_rows(1, 0, 100000, 128); Q = oracle.uniform_rows(2, 0, 40000, 128)
h = oracle.HNSW(16, 200)
for r in V: h.insert(r, 0)
for T in (1, 2, 4, 8, 16, 32):
    t0=time.perf_counter(); h.search_many(Q, 10, threads=T); el=time.perf_counter()-t0; print(T, "threads", round(len(Q)/el), "QPS", flush=True)
