# certificate cost probe: time per batch and candidate rows evaluated by the re-rank at each survivor level, on the benchmark's generator
# usage: python tools/cert_probe.py [rows] [dim]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zvdb_amd as zv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
B, K = 1024, 10
dev = torch.device("cuda", 0); s = torch.cuda.current_stream().cuda_stream
idx = zv.Index(dim=D, capacity=N, flags=zv.FLAG_BF16_MIRROR)
idx.add_synthetic(N, D, 0x5EED)
q = torch.empty((8, B, D), dtype=torch.float32, device=dev)
zv.fill_synthetic_dev(q.data_ptr(), 8 * B, D, 0xC0FFEE, 0, s)
wsb = idx.bruteforce_workspace(B, K, zv.BF_BF16); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)
zv.set_tuning("ZVDB_CERT_DEBUG", sys.argv[3] if len(sys.argv) > 3 else "1")  # 1: rows evaluated per query; 2: phase stamps of every 256th query
def run(n):
    t0 = time.perf_counter()
    for i in range(n):
        idx.search_bruteforce_dev(q[i % 8].data_ptr(), B, D, K, zv.BF_BF16, 0, ws.data_ptr(), wsb, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
idx.set_option(zv.OPT_CERTIFY, 0); run(3); print(f"uncertified: {run(10):.3f} ms per batch", flush=True)
idx.set_option(zv.OPT_CERTIFY, 1)
for lvl in (3, 2, 1):
    idx.set_option(zv.OPT_CERT_MIN_LEVEL, lvl)
    a = idx.cert_info(); run(3); b = idx.cert_info(); ms = run(10); c = idx.cert_info()
    print(f"level >= {lvl}: {ms:.3f} ms per batch; re-rank rows per query {(c['rerank_rows'] - b['rerank_rows']) / (10 * B):.1f}; repaired in 10 batches {c['repaired'] - b['repaired']} (level now {c['level']}); re-rank phases (query constants, first batch, further candidates, final bound, ranking + output), shader cycles per sampled query: {[round((x - y) / 40.0) for x, y in zip(c['rerank_cycles'], b['rerank_cycles'])]}", flush=True)
