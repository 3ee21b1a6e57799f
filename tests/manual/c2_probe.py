"""BASELINE configs[1] (1M x 128 fp32 brute force, batch 256) on every exact screen: QPS with resident queries + oracle spot check."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, oracle, zvdb_amd as zv
N, D, B, K = 1_000_000, 128, 256, 10
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
idx = zv.Index(dim=D, capacity=N, flags=zv.FLAG_BF16_MIRROR); idx.add_synthetic(N, D, 0x5EED)
q = torch.empty((B, D), dtype=torch.float32, device=dev); zv.fill_synthetic_dev(q.data_ptr(), B, D, 0xC0FFEE, 0, stream)
ws_b = max(idx.bruteforce_workspace(B, K, m) for m in (zv.BF_F32, zv.BF_BF16, zv.BF_I8, zv.BF_FP4)); ws = torch.empty(ws_b, dtype=torch.uint8, device=dev)
ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)
V = oracle.uniform_rows(0x5EED, 0, N, D); ri, rd, _ = oracle.bruteforce_topk(V, q.cpu().numpy()[:4], K)
for name, m in (("f32", zv.BF_F32), ("bf16", zv.BF_BF16), ("i8", zv.BF_I8), ("fp4", zv.BF_FP4)):
    for _ in range(3): idx.search_bruteforce_dev(q.data_ptr(), B, D, K, m, 0, ws.data_ptr(), ws_b, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): idx.search_bruteforce_dev(q.data_ptr(), B, D, K, m, 0, ws.data_ptr(), ws_b, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), stream)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 50
    ok = np.array_equal(ids.cpu().numpy()[:4].astype(np.uint64), ri) and np.array_equal(dd.cpu().numpy()[:4], rd)
    print(f"{name}: {B / el:10.0f} QPS  {el * 1e6:7.1f} us per batch  {2.0 * B * N * D / el / 1e12:7.1f} TFLOP/s  exact vs oracle (4 queries): {ok}", flush=True)
