# K2 (reference-compat search, hnsw.zig:194-236) with RESIDENT queries: kernel-only throughput vs its HBM-gather roofline
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, oracle, zvdb_amd as zv
from zvdb_amd import _lib
from zvdb_amd.index import Index
for P, D in ((100_000, 128), (100_000, 768)):
    V = oracle.uniform_rows(1, 0, P, D)
    h = zv.HNSW(16, 200)
    t0 = time.perf_counter()
    for r in V: h.insert(r, _level=0)
    t_ins = time.perf_counter() - t0
    idx = Index._from_handle(h._L.zvdb_host_hnsw_handle(h._p))
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    for NQ in (10_000, 200_000):
        K = 10
        q = torch.empty((NQ, D), dtype=torch.float32, device=dev)
        zv.fill_synthetic_dev(q.data_ptr(), NQ, D, 0xC0FFEE, 0, s)
        ids = torch.empty((NQ, K), dtype=torch.int64, device=dev); dd = torch.empty((NQ, K), dtype=torch.float32, device=dev)
        cnt = torch.empty((NQ,), dtype=torch.int32, device=dev); st = torch.zeros(8, dtype=torch.int64, device=dev)
        idx.search_graph_dev(q.data_ptr(), NQ, D, K, 0, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), s)
        torch.cuda.synchronize()
        ev, pp = st[0].item(), st[1].item()
        t0 = time.perf_counter()
        for _ in range(5):
            idx.search_graph_dev(q.data_ptr(), NQ, D, K, 0, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 5
        by = ev * D * 4 + pp * 16 * 4
        print(f"{P}x{D} (insert {P / t_ins:,.0f} pts/s): nq {NQ}: {el * 1e3:.3f} ms = {NQ / el:,.0f} QPS, {ev / NQ:.1f} evals/q, "
              f"{by / el / 1e9:,.0f} GB/s algorithmic = {by / el / 8e12:.3f} of 8 TB/s", flush=True)
