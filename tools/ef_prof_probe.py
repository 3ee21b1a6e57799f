# per-phase shader cycles of the register-beam search (the PROF instantiation: one wave per query, width 1) at several ef
# usage: python tools/ef_prof_probe.py [ef ...]     (prints the library's "[zvdb ef prof]" lines)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zvdb_amd as zv, bench
N, D, B, K, M = 1_000_000, 768, 1024, 10, 16
EFS = [int(x) for x in sys.argv[1:]] or [128, 256, 512]
dev = torch.device("cuda", 0); s = torch.cuda.current_stream().cuda_stream
for hb in (True, False):
    g = zv.Index(dim=D, capacity=N, m=M, graph_stride=M, flags=zv.FLAG_BF16_MIRROR | (zv.FLAG_BF16_ROWS if hb else 0))
    q = bench.make_rows(zv, torch, g, N, D, False, dev, s, B)
    g.build_knn_graph(M, zv.BF_I8)
    ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    for ef in EFS:
        for _ in range(2): g.search_graph_dev(q.data_ptr(), B, D, K, ef, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
        torch.cuda.synchronize()
        print("bf16 rows" if hb else "fp32 rows", "ef", ef, flush=True)
        with zv.tuned(ZVDB_EF_PROF="1"):
            g.search_graph_dev(q.data_ptr(), B, D, K, ef, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
        torch.cuda.synchronize()
    del g
