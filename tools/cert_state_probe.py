# does a batch at a lower survivor level / an exact-scan repair leave state behind that slows the following scans?
# (one process, fresh index per leg; usage: python tools/cert_state_probe.py)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zvdb_amd as zv
N, D, B, K = 10_000_000, 768, 1024, 10
dev = torch.device("cuda", 0); s = torch.cuda.current_stream().cuda_stream
q = torch.empty((8, B, D), dtype=torch.float32, device=dev); zv.fill_synthetic_dev(q.data_ptr(), 8 * B, D, 0xC0FFEE, 0, s)
ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)
def leg(tag, prep):
    idx = zv.Index(dim=D, capacity=N, flags=zv.FLAG_BF16_MIRROR); idx.add_synthetic(N, D, 0x5EED)
    wsb = max(idx.bruteforce_workspace(B, K, m) for m in (zv.BF_BF16, zv.BF_F32)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    def run(n, mode=zv.BF_BF16):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n): idx.search_bruteforce_dev(q[i % 8].data_ptr(), B, D, K, mode, 0, ws.data_ptr(), wsb, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    prep(idx, run)
    run(5)
    a, b = run(20), run(20)
    ci = idx.cert_info()
    print(f"{tag}: {a:.3f} {b:.3f} ms per batch; level {ci['level']}, repaired {ci['repaired']}", flush=True)
    del idx, ws
leg("level 2 from the start", lambda idx, run: idx.set_option(zv.OPT_CERT_MIN_LEVEL, 2))
leg("auto (one fp32 batch, then bf16)", lambda idx, run: (run(1, zv.BF_F32), run(1)))
leg("level 2 from the start, again", lambda idx, run: idx.set_option(zv.OPT_CERT_MIN_LEVEL, 2))
leg("auto, bf16 only", lambda idx, run: run(1))
leg("level 2 + one fp32 batch first", lambda idx, run: (idx.set_option(zv.OPT_CERT_MIN_LEVEL, 2), run(1, zv.BF_F32)))
leg("uncertified", lambda idx, run: idx.set_option(zv.OPT_CERTIFY, 0))
