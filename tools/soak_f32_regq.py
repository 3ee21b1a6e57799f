# soak: many different query batches through the fp32 register-query scan (bf_f32_regq.hip; thresholds from the rows' moments) on four row widths (every ring geometry up to 256 dimensions but 28 / 32 pieces);
# every batch compared with the certified int8 route bit for bit (both return the exact top-k), certificate counters printed at the end.
# usage: python tools/soak_f32_regq.py [batches per shape]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zvdb_amd as zv
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
bad = 0
t0 = time.time()
for N, D, B, K in ((2_000_000, 128, 300, 10), (1_000_000, 200, 1024, 32), (1_500_000, 64, 257, 100), (1_200_000, 96, 300, 10)):
    idx = zv.Index(dim=D, capacity=N)
    idx.add_synthetic(N, D, 0x5EED)
    wsb = max(idx.bruteforce_workspace(B, K, m) for m in (zv.BF_F32, zv.BF_I8))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    q = torch.empty((B, D), dtype=torch.float32, device=dev)
    out = {m: (torch.empty((B, K), dtype=torch.int64, device=dev), torch.empty((B, K), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev))
           for m in (zv.BF_F32, zv.BF_I8)}
    for b in range(NB):
        zv.fill_synthetic_dev(q.data_ptr(), B, D, 0xC0FFEE + 7919 * b, 0, s)
        for m in (zv.BF_F32, zv.BF_I8):
            i, d, c = out[m]
            idx.search_bruteforce_dev(q.data_ptr(), B, D, K, m, 0, ws.data_ptr(), wsb, i.data_ptr(), d.data_ptr(), c.data_ptr(), s)
        torch.cuda.synchronize()
        a, r = out[zv.BF_F32], out[zv.BF_I8]
        ok = bool(torch.equal(a[0], r[0]) and torch.equal(a[1], r[1]) and torch.equal(a[2], r[2]) and bool((a[2] == K).all()))
        if not ok:
            bad += 1
            print(f"{N}x{D} batch {b}: fp32 register-query route != int8 route", flush=True)
        if b % 250 == 249: print(f"{N}x{D} b{B} k{K}: {b + 1} batches, {bad} bad, {time.time() - t0:.0f} s", flush=True)
    print(f"{N}x{D}: certificate {idx.cert_info()}", flush=True)
print("done:", bad, "bad")
