# recall / speed of the approximate (fp4) and exact (int8) screens on NON-uniform data: Gaussian and clustered rows
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, zvdb_amd as zv
rng = np.random.default_rng(7)
N, D, NQ, K = 1_000_000, 128, 1024, 10
def run(name, V, Q):
    idx = zv.Index()
    for i in range(0, N, 250_000): idx.add(V[i:i + 250_000])
    ei, ed, _ = idx.search_bruteforce(Q, K, zv.BF_F32)
    for mode, mname in ((zv.BF_FP4, "fp4"), (zv.BF_I8, "i8")):
        idx.search_bruteforce(Q, K, mode)                      # builds the mirror, adapts the guess
        for _ in range(3): idx.search_bruteforce(Q, K, mode)
        t0 = time.perf_counter()
        for _ in range(5): ai, ad, _ = idx.search_bruteforce(Q, K, mode)
        dt = (time.perf_counter() - t0) / 5
        rec = np.mean([len(set(ai[i]) & set(ei[i])) / K for i in range(NQ)])
        print(f"{name:10s} {mname:4s} recall@10 {rec:.4f}  {dt * 1e3:.2f} ms per 1024-query batch (host API, PCIe included)")
V = rng.standard_normal((N, D), dtype=np.float32); Q = rng.standard_normal((NQ, D), dtype=np.float32)
run("gaussian", V, Q)
C = rng.standard_normal((64, D), dtype=np.float32) * 3
V = (C[rng.integers(0, 64, N)] + rng.standard_normal((N, D), dtype=np.float32)).astype(np.float32)
Q = (C[rng.integers(0, 64, NQ)] + rng.standard_normal((NQ, D), dtype=np.float32)).astype(np.float32)
run("clustered", V, Q)
# more fp4 survivors on the clustered rows (zvdb_hip_index_set_screen_survivors)
idx = zv.Index()
for i in range(0, N, 250_000): idx.add(V[i:i + 250_000])
ei, _, _ = idx.search_bruteforce(Q, K, zv.BF_F32)
for sv in (64, 128, 256):
    idx.set_screen_survivors(sv)
    for _ in range(6): idx.search_bruteforce(Q, K, zv.BF_FP4)
    t0 = time.perf_counter()
    for _ in range(5): ai, _, _ = idx.search_bruteforce(Q, K, zv.BF_FP4)
    dt = (time.perf_counter() - t0) / 5
    rec = np.mean([len(set(ai[i]) & set(ei[i])) / K for i in range(NQ)])
    print(f"clustered  fp4 survivors {sv}: recall@10 {rec:.4f}  {dt * 1e3:.2f} ms")
