# soak: many different query batches through the guessed-threshold pipelines; every batch checked (fp4 recall vs the int8 route,
# the bf16 stream route == the int8 route, int8 vs the fp32 route on sampled batches, counts, ascending distances)
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zvdb_amd as zv
N, D, B, K = 2_000_000, 768, 1024, 10
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
idx = zv.Index(dim=D, capacity=N, flags=zv.FLAG_BF16_MIRROR)
idx.add_synthetic(N, D, 0x5EED)
s = torch.cuda.current_stream().cuda_stream
wsb = max(idx.bruteforce_workspace(B, K, m) for m in (zv.BF_F32, zv.BF_BF16, zv.BF_I8, zv.BF_FP4))
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
q = torch.empty((B, D), dtype=torch.float32, device=dev)
def run(mode):
    ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    idx.search_bruteforce_dev(q.data_ptr(), B, D, K, mode, 0, ws.data_ptr(), wsb, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s)
    torch.cuda.synchronize()
    return ids.cpu().numpy(), dd.cpu().numpy(), cnt.cpu().numpy()
worst, bad = 1.0, 0
t0 = time.time()
for b in range(NB):
    zv.fill_synthetic_dev(q.data_ptr(), B, D, 0xC0FFEE + 7919 * b, 0, s)
    i4, d4, c4 = run(zv.BF_FP4)
    i8, d8, c8 = run(zv.BF_I8)
    ib, db, cb = run(zv.BF_BF16)
    ok = bool(np.array_equal(ib, i8) and np.array_equal(db, d8) and np.array_equal(cb, c8) and np.all(c4 == K) and np.all(c8 == K) and np.all(np.diff(d4, axis=1) >= 0) and np.all(np.diff(d8, axis=1) >= 0))
    rec = float(np.mean([len(set(i4[i]) & set(i8[i])) / K for i in range(B)]))
    worst = min(worst, rec)
    if b % 25 == 0:
        i32, d32, _ = run(zv.BF_F32)
        ok = ok and np.array_equal(i32, i8) and np.array_equal(d32, d8)
    if not ok or rec < 0.985:
        bad += 1
        print(f"batch {b}: ok={ok} fp4 recall {rec:.4f}", flush=True)
    if b % 50 == 49: print(f"{b + 1} batches, worst fp4 recall {worst:.4f}, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("done:", NB, "batches, worst fp4 recall", round(worst, 4), "bad", bad)
