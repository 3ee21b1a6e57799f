# one-off randomized parity sweep of the brute-force routes against the oracle (exact screens: ids + distances bit for bit;
# fp4: counts, order, exact distances of whatever it returns).  usage: python tests/manual/fuzz_bruteforce.py [cases] [seed]
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle, zvdb_amd as zv
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    n = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 5000, 20000, 70000]))
    n = max(1, n + int(rng.integers(-3, 4)))
    d = int(rng.choice([1, 2, 3, 7, 16, 31, 32, 33, 64, 96, 100, 128, 129, 200, 256, 300, 384, 512, 768, 769, 1024]))
    nq = int(rng.choice([1, 2, 37, 64, 128, 129, 255, 256, 257, 300]))
    k = int(rng.choice([1, 2, 5, 10, 16, 17, 32, 33, 48, 49, 56, 57, 64, 100, 112]))
    metric = int(rng.choice([0, 0, 0, 1]))
    if n * d * nq > 4e9: nq = max(1, int(4e9 / (n * d)))
    V = oracle.uniform_rows(1000 + c, 0, n, d) - (0.5 if metric else 0.0)
    Q = oracle.uniform_rows(5000 + c, 0, nq, d) - (0.5 if metric else 0.0)
    idx = zv.Index(metric=zv.METRIC_DOT if metric else zv.METRIC_L2SQ, flags=zv.FLAG_BF16_MIRROR)
    half = n // 2
    if half: idx.add(V[:half])
    idx.add(V[half:])
    ri, rd, rc = oracle.bruteforce_topk(V, Q, k, metric=metric)
    tag = f"case {c}: n={n} d={d} nq={nq} k={k} metric={metric}"
    for mode, name in ((zv.BF_F32, "f32"), (zv.BF_BF16, "bf16"), (zv.BF_I8, "i8")):
        ids, dist, cnt = idx.search_bruteforce(Q, k, mode)
        ok = np.array_equal(cnt, rc)
        exact = True
        for qi in range(nq):
            cq = int(rc[qi])
            if not (np.array_equal(ids[qi, :cq], ri[qi, :cq]) and np.array_equal(dist[qi, :cq], rd[qi, :cq])):
                exact = False
                break
        if metric and name == "i8":  # (8-bit screen under the inner product: near-exact, not guaranteed)
            rec = np.mean([len(set(ids[q][:int(rc[q])]) & set(ri[q][:int(rc[q])])) / max(1, int(rc[q])) for q in range(nq)])
            exact = rec >= 0.97
        if not (ok and exact):
            bad += 1
            print("MISMATCH", tag, name, flush=True)
    a4, d4, c4 = idx.search_bruteforce(Q, k, zv.BF_FP4)
    ok4 = np.array_equal(c4, rc)
    for qi in range(0, nq, max(1, nq // 5)):
        cq = int(rc[qi])
        ok4 = ok4 and bool(np.all(np.diff(d4[qi, :cq]) >= 0))
        for j in range(0, cq, max(1, cq // 4)):
            want = oracle.distance(Q[qi], V[int(a4[qi, j])]) if not metric else None
            if want is not None and want != d4[qi, j]: ok4 = False
    if not ok4:
        bad += 1
        print("MISMATCH", tag, "fp4", flush=True)
    if c % 10 == 9: print(f"{c + 1} cases, {bad} mismatches", flush=True)
print("done:", cases, "cases,", bad, "mismatches")
