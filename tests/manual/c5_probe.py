# C5-shaped sanity (BASELINE configs[4]: 1536-d, cosine, bf16 storage, recall@100) at a size the oracle can check
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, oracle, zvdb_amd as zv
for n in (30_000, 300_000):
    d, nq, k = 1536, 16, 100
    V = oracle.uniform_rows(31, 0, n, d) - 0.5
    Q = oracle.uniform_rows(32, 0, nq, d) - 0.5
    idx = zv.Index(metric=zv.METRIC_COSINE, flags=zv.FLAG_BF16_MIRROR | zv.FLAG_BF16_ROWS)
    idx.add(V)
    stored = np.stack([idx.get_vector(i) for i in range(0, n, 997)])
    assert not np.any(stored.view(np.uint32) & 0xFFFF)
    t0 = time.time()
    ai, ad, ac = idx.search_bruteforce(Q, k, zv.BF_FP4)
    t1 = time.time()
    # ground truth: oracle over the stored (normalised, rounded) rows
    S = np.empty((n, d), dtype=np.float32)
    for i in range(n):
        S[i] = idx.get_vector(i)
    ri, rd, _ = oracle.bruteforce_topk(S, Q, k, metric=1)
    rec = np.mean([len(set(ai[i]) & set(ri[i])) / k for i in range(nq)])
    print(f"n={n}: fp4 k=100 recall@100 {rec:.4f}, counts ok {bool(np.all(ac == k))}, sorted {bool(np.all(np.diff(ad, axis=1) >= 0))}, search {t1 - t0:.3f}s", flush=True)
