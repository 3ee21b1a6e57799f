import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, oracle, zvdb_amd as zv
from zvdb_amd.index import Index
P, D = 100_000, 128
V = oracle.uniform_rows(1, 0, P, D)
h = zv.HNSW(16, 200)
h.insert_many(V, np.zeros(P, dtype=np.int32))
idx = Index._from_handle(h._L.zvdb_host_hnsw_handle(h._p))
dev = torch.device("cuda", 0); s = torch.cuda.current_stream().cuda_stream
for K in (10, 25, 50, 100):
    NQ = 50_000
    q = torch.empty((NQ, D), dtype=torch.float32, device=dev)
    zv.fill_synthetic_dev(q.data_ptr(), NQ, D, 0xC0FFEE, 0, s)
    ids = torch.empty((NQ, K), dtype=torch.int64, device=dev); dd = torch.empty((NQ, K), dtype=torch.float32, device=dev)
    cnt = torch.empty((NQ,), dtype=torch.int32, device=dev)
    idx.search_graph_dev(q.data_ptr(), NQ, D, K, 0, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        idx.search_graph_dev(q.data_ptr(), NQ, D, K, 0, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 3
    print(f"k={K}: {NQ / el:,.0f} QPS", flush=True)
