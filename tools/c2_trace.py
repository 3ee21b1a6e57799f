import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zvdb_amd as zv
N2, D2, B2, K = 1_000_000, (int(sys.argv[2]) if len(sys.argv) > 2 else 128), 256, 10   # c2_trace.py [iterations [dim]]
dev = torch.device("cuda", 0)
idx = zv.Index(dim=D2, capacity=N2, flags=zv.FLAG_BF16_MIRROR)
idx.add_synthetic(N2, D2, 0x5EED)
q = torch.empty((B2, D2), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
zv.fill_synthetic_dev(q.data_ptr(), B2, D2, 0xC0FFEE, 0, s)
wsb = idx.bruteforce_workspace(B2, K, zv.BF_F32)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
ids = torch.empty((B2, K), dtype=torch.int64, device=dev); dd = torch.empty((B2, K), dtype=torch.float32, device=dev); cnt = torch.empty((B2,), dtype=torch.int32, device=dev)
import contextlib
tune = dict(kv.split("=") for kv in os.environ.get("C2_TUNE", "").split(",") if "=" in kv)   # C2_TUNE=ZVDB_RERANK=v1,... : tuning switches for the run
with (zv.tuned(**tune) if tune else contextlib.nullcontext()):
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        idx.search_bruteforce_dev(q.data_ptr(), B2, D2, K, zv.BF_F32, 0, ws.data_ptr(), wsb, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s)
    torch.cuda.synchronize()
