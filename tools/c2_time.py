import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zvdb_amd as zv, numpy as np
N2, D2, B2, K = (int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000), int(sys.argv[1]) if len(sys.argv) > 1 else 128, 256, 10
dev = torch.device("cuda", 0)
idx = zv.Index(dim=D2, capacity=N2)
idx.add_synthetic(N2, D2, 0x5EED)
q = torch.empty((B2, D2), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
zv.fill_synthetic_dev(q.data_ptr(), B2, D2, 0xC0FFEE, 0, s)
wsb = idx.bruteforce_workspace(B2, K, zv.BF_F32)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
ids = torch.empty((B2, K), dtype=torch.int64, device=dev); dd = torch.empty((B2, K), dtype=torch.float32, device=dev); cnt = torch.empty((B2,), dtype=torch.int32, device=dev)
for _ in range(5):
    idx.search_bruteforce_dev(q.data_ptr(), B2, D2, K, zv.BF_F32, 0, ws.data_ptr(), wsb, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    idx.search_bruteforce_dev(q.data_ptr(), B2, D2, K, zv.BF_F32, 0, ws.data_ptr(), wsb, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / 50
print(f"dim {D2}: {el * 1e3:.3f} ms per batch, {B2 / el:.0f} QPS, {2.0 * B2 * N2 * D2 / el / 1e12:.1f} TFLOP/s; checksum {int(ids.sum().item())} {float(dd.sum().item()):.3f}")
