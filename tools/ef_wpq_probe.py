import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch, zvdb_amd as zv, bench
N, D, B, K, M = 1_000_000, 768, 1024, 10, 16
dev = torch.device("cuda", 0); s = torch.cuda.current_stream().cuda_stream
g = zv.Index(dim=D, capacity=N, m=32, graph_stride=32, flags=zv.FLAG_BF16_MIRROR | zv.FLAG_BF16_ROWS)
q = bench.make_rows(zv, torch, g, N, D, True, dev, s, B)
g.build_graph(M, 64, zv.BF_I8, 7)
ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)
for ef in (128, 256):
  for width in (1, 2, 4):
    for wpq in ("1", "2", "4"):
        g.set_option(zv.OPT_EF_SEARCH_WIDTH, width); zv.set_tuning("ZVDB_EF_WPQ", wpq)
        for _ in range(3): g.search_graph_dev(q.data_ptr(), B, D, K, ef, zv.ENTRY_DESCEND, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
        torch.cuda.synchronize(); g.timing_enable(True)
        for _ in range(8): g.search_graph_dev(q.data_ptr(), B, D, K, ef, zv.ENTRY_DESCEND, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
        torch.cuda.synchronize(); kms, kn = g.timing_read(); g.timing_enable(False)
        print(f"ef {ef} width {width} wpq {wpq}: kernel {kms / kn:.3f} ms", flush=True)
