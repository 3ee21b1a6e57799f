# randomized parity sweep of the certificate's repair pass ALONE (ZVDB_OPT_CERT_FORCE_REPAIR: every query of every batch is recomputed by
# k_exact_scan, exact_scan.hip) against the oracle's brute force: ids, distances (bit patterns, incl. -0.0) and counts.  Shapes: ragged
# row counts and dimensions, 1 .. 100 queries (lists of 8 / 16 / 32 sums per pass, several passes, several workers), k up to 240 and
# beyond the row count (no start bound: the flood + prune path), rows with many exact duplicates (ties at the k-th distance), every
# metric, fp32 and bf16 row storage.  usage: python tests/manual/fuzz_exact_scan.py [cases] [seed]
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle, zvdb_amd as zv
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    n = max(1, int(rng.choice([1, 3, 64, 511, 512, 513, 1025, 4000, 20000, 60000])) + int(rng.integers(-2, 3)))
    d = int(rng.choice([1, 3, 8, 31, 32, 33, 64, 65, 96, 128, 200, 256, 257, 512, 768]))
    nq = int(rng.choice([1, 2, 7, 8, 9, 16, 17, 32, 33, 70, 100]))
    k = int(rng.choice([1, 2, 10, 16, 31, 32, 33, 64, 100, 128, 129, 240]))
    metric = str(rng.choice(["l2", "l2", "dot", "cosine"]))
    bf16rows = bool(rng.integers(0, 3) == 0)
    dup = bool(rng.integers(0, 3) == 0)
    if n * d * nq > 3e8: nq = max(1, int(3e8 / (n * d)))
    if dup:  # few distinct rows, small integer coordinates: exact ties everywhere, also at the k-th distance
        base = rng.integers(-2, 3, size=(max(1, n // 7), d)).astype(np.float32)
        V = base[rng.integers(0, len(base), size=n)]
        Q = rng.integers(-2, 3, size=(nq, d)).astype(np.float32)
    else:
        V = oracle.uniform_rows(3000 + c, 0, n, d) - (0.0 if metric == "l2" else 0.5)
        Q = oracle.uniform_rows(7000 + c, 0, nq, d) - (0.0 if metric == "l2" else 0.5)
    if metric == "cosine": V = V + np.float32(0.01)  # (no zero rows under cosine: the oracle normalises at insert)
    mt = {"l2": zv.METRIC_L2SQ, "dot": zv.METRIC_DOT, "cosine": zv.METRIC_COSINE}[metric]
    idx = zv.Index(dim=d, metric=mt, flags=zv.FLAG_BF16_ROWS if bf16rows else zv.FLAG_BF16_MIRROR)
    idx.add(V)
    idx.set_option(zv.OPT_CERT_FORCE_REPAIR, 1)
    Vs = np.stack([idx.get_vector(i) for i in range(n)]) if (metric == "cosine" or bf16rows) else V  # rows as stored
    ri, rd, rc = oracle.bruteforce_topk(Vs, Q, k, 0 if metric == "l2" else 1)
    tag = f"case {c}: n={n} d={d} nq={nq} k={k} metric={metric} bf16rows={bf16rows} dup={dup}"
    ran = 0
    for mode, name in ((zv.BF_F32, "f32"), (zv.BF_BF16, "bf16"), (zv.BF_I8, "i8")):
        if k > 112 and mode == zv.BF_F32 and d > 128: continue  # (k <= 112 on the staged fp32 route)
        try:
            ids, dist, cnt = idx.search_bruteforce(Q, k, mode)
        except zv.ZvdbError as e:
            if "supports k <=" in str(e): continue   # (k beyond what the screen route of this shape keeps: no search, nothing to repair)
            raise
        ran += 1
        ok = np.array_equal(cnt, rc)
        for qi in range(nq):
            cq = int(rc[qi])
            ok = ok and np.array_equal(ids[qi, :cq], ri[qi, :cq]) and np.array_equal(dist[qi, :cq].view(np.uint32), rd[qi, :cq].view(np.uint32))
            ok = ok and bool(np.all(ids[qi, cq:] == np.uint64(0xFFFFFFFFFFFFFFFF)))
        if not ok:
            bad += 1
            print("MISMATCH", tag, name, flush=True)
    if idx.cert_info()["repaired"] < nq * ran:
        bad += 1
        print("NOT REPAIRED", tag, idx.cert_info(), flush=True)
    if c % 10 == 9: print(f"{c + 1} cases, {bad} mismatches", flush=True)
print("done:", cases, "cases,", bad, "mismatches")
