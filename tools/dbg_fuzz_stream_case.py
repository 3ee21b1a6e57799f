import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, oracle, zvdb_amd as zv
seed0, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0)
for c in range(target + 1):
    d = int(rng.choice([64, 96, 100, 128, 192, 200, 384, 768]))
    n = int(rng.choice([262144, 262145, 300001, 500000, 777777, 1000003]))
    n = min(n, int(1.0e8 // d))
    nq = int(rng.choice([1, 255, 256, 257, 700, 1024, 1100]))
    k = int(rng.choice([1, 10, 16, 17, 32, 48, 64, 100, 112, 200, 240]))
    metric = int(rng.choice([0, 0, 0, 1]))
    seed = 7000 + c
    first = int(rng.integers(1, n))
print("case", c, dict(d=d, n=n, nq=nq, k=k, metric=metric, first=first), flush=True)
idx = zv.Index(metric=zv.METRIC_DOT if metric else zv.METRIC_L2SQ, flags=zv.FLAG_BF16_MIRROR)
idx.add_synthetic(first, d, seed)
idx.add_synthetic(n - first, d, seed, first_row=first)
V = oracle.uniform_rows(seed, 0, n, d)
Q = oracle.uniform_rows(seed + 100000, 0, nq, d)
sel = list(range(0, nq, max(1, nq // 24)))
ri, rd, _ = oracle.bruteforce_topk(V, Q[sel], k, metric=metric)
for name, mode in (("i8", zv.BF_I8), ("bf16", zv.BF_BF16), ("f32", zv.BF_F32) if k <= 56 else ("i8b", zv.BF_I8), ("fp4", zv.BF_FP4)):
    for rep in range(3):
        ii, dd, cc = idx.search_bruteforce(Q, k, mode)
        nbad = 0; worst = None
        for j, qi in enumerate(sel):
            same_i = np.array_equal(ii[qi], ri[j]); same_d = np.array_equal(dd[qi], rd[j])
            if not (same_i and same_d):
                nbad += 1
                if worst is None:
                    miss = sorted(set(ri[j].tolist()) - set(ii[qi].tolist()))
                    worst = (qi, len(miss), miss[:5], [int(np.where(ri[j] == m)[0][0]) for m in miss[:5]])
        asc = bool(np.all(np.diff(dd, axis=1) >= 0))
        print(name, "rep", rep, "counts ok", bool(np.all(cc == k)), "ascending", asc, "queries differing from oracle", nbad, "of", len(sel), "first:", worst, flush=True)
