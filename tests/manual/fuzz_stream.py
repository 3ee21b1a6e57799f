# one-off randomized sweep of the LARGE-shard brute-force pipelines (stream kernels, guessed + verified thresholds, wide k)
# against the oracle on sampled queries.  usage: python tests/manual/fuzz_stream.py [cases] [seed]
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle, zvdb_amd as zv
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    d = int(rng.choice([64, 96, 100, 128, 160, 192, 200, 256, 384, 768]))
    n = int(rng.choice([262144, 262145, 300001, 500000, 777777, 1000003]))
    n = min(n, int(1.0e8 // d))
    nq = int(rng.choice([1, 255, 256, 257, 700, 1024, 1100]))
    k = int(rng.choice([1, 10, 16, 17, 32, 48, 64, 100, 112, 200, 240]))
    metric = int(rng.choice([0, 0, 0, 1]))
    seed = 7000 + c
    idx = zv.Index(metric=zv.METRIC_DOT if metric else zv.METRIC_L2SQ, flags=zv.FLAG_BF16_MIRROR)
    first = int(rng.integers(1, n))
    idx.add_synthetic(first, d, seed)                       # two adds: the second one ragged
    idx.add_synthetic(n - first, d, seed, first_row=first)
    V = oracle.uniform_rows(seed, 0, n, d)
    Q = oracle.uniform_rows(seed + 100000, 0, nq, d)
    sel = sorted(set([0, nq // 3, nq - 1]))
    ri, rd, _ = oracle.bruteforce_topk(V, Q[sel], k, metric=metric)
    tag = f"case {c}: n={n} d={d} nq={nq} k={k} metric={metric}"
    ok = True
    i8, d8, c8 = idx.search_bruteforce(Q, k, zv.BF_I8)
    if not np.all(c8 == k): ok = False
    for j, qi in enumerate(sel):
        if metric == 0 and k <= 200:
            if not (np.array_equal(i8[qi], ri[j]) and np.array_equal(d8[qi], rd[j])): ok = False
        elif metric == 0:
            # k = 240 of k' = 256 survivors: the int8 screen's margin is 16 keys -- a neighbour of rank ~236 can fall outside it
            # (seed 231, case 16: 777,777 x 128, one query in 25 lost its 236th neighbour).  The screen is exact "whenever the true
            # top-k are among its k' best" (include/zvdb_hip.h); what is held here: >= 99 % of the ids, every returned distance exact
            if len(set(i8[qi]) & set(ri[j])) / k < 0.99: ok = False
            for e in range(0, k, max(1, k // 8)):
                if oracle.distance(Q[qi], V[int(i8[qi, e])]) != d8[qi, e]: ok = False
        else:
            if len(set(i8[qi]) & set(ri[j])) / k < 0.95: ok = False
    if d <= 768 and k <= 240:  # the bf16 register-resident-query stream kernel (exact screen under L2)
        b16, db16, cb16 = idx.search_bruteforce(Q, k, zv.BF_BF16)
        if not np.all(cb16 == k): ok = False
        for j, qi in enumerate(sel):
            if metric == 0:
                if not (np.array_equal(b16[qi], ri[j]) and np.array_equal(db16[qi], rd[j])): ok = False
            elif len(set(b16[qi]) & set(ri[j])) / k < 0.95: ok = False
    if k <= 112 or d <= 128 or (d <= 256 and nq > 128):  # the fp32 screen: register-query scan (dim <= 256, batches > 128), 128-query stream kernel, staged kernel
        f32, df32, cf32 = idx.search_bruteforce(Q, k, zv.BF_F32)
        if not np.all(cf32 == k): ok = False
        for j, qi in enumerate(sel):
            if metric == 0:
                if not (np.array_equal(f32[qi], ri[j]) and np.array_equal(df32[qi], rd[j])): ok = False
            elif len(set(f32[qi]) & set(ri[j])) / k < 0.95: ok = False
    a4, d4, c4 = idx.search_bruteforce(Q, k, zv.BF_FP4)
    if not (np.all(c4 == k) and np.all(np.diff(d4, axis=1) >= 0)): ok = False
    rec = np.mean([len(set(a4[qi]) & set(ri[j])) / k for j, qi in enumerate(sel)])
    if rec < 0.6: ok = False
    if metric == 0:
        for j, qi in enumerate(sel):
            for e in range(0, k, max(1, k // 5)):
                if oracle.distance(Q[qi], V[int(a4[qi, e])]) != d4[qi, e]: ok = False
    print(("ok      " if ok else "MISMATCH") + " " + tag + f" fp4 recall {rec:.3f}", flush=True)
    bad += 0 if ok else 1
print("done:", cases, "cases,", bad, "mismatches")
