# first call of an ef bucket against the later ones (visited-table sizing before any fill has been seen): python tools/ef_first_call.py
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zvdb_amd as zv
N, d, M = 300000, 1536, 16
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
idx = zv.Index(dim=d, capacity=N, m=2*M, graph_stride=2*M, metric=zv.METRIC_COSINE, flags=zv.FLAG_BF16_ROWS)
g = torch.Generator(device=dev); g.manual_seed(5)
A = torch.rand((12, d), generator=g, device=dev) - 0.5
rows = (torch.rand((N, 12), generator=g, device=dev) @ A + 0.01 * torch.randn((N, d), generator=g, device=dev)).contiguous()
torch.cuda.synchronize(); idx.add_dev(rows.data_ptr(), N, d, stream); torch.cuda.synchronize()
idx.build_graph(M, 64, zv.BF_I8, 7)
nq, k = 1024, 100
q = (torch.rand((nq, 12), generator=g, device=dev) @ A).contiguous()
ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dd = torch.empty((nq, k), dtype=torch.float32, device=dev); cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for ef in (128, 256):
    ts = []
    for r in range(4):
        t0 = time.perf_counter()
        idx.search_graph_dev(q.data_ptr(), nq, d, k, ef, zv.ENTRY_DESCEND, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, stream)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("ef", ef, "ms per call:", [round(t, 2) for t in ts], int(ids.sum().item()))
