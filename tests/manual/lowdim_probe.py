# exactness of the "exact" screens on dense low-dimensional data (many rows inside the screen's rounding band)
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, oracle, zvdb_amd as zv
cases = [(69998, 1, 255, 1), (70000, 2, 256, 10), (200000, 1, 100, 10), (200000, 3, 100, 10), (1000000, 4, 64, 10),
         (300000, 5, 64, 10), (300000, 6, 64, 10), (300000, 8, 64, 10), (1000000, 8, 32, 48), (300000, 12, 64, 10), (300000, 16, 64, 10),
         (2000000, 6, 16, 10)]
for c, (n, d, nq, k) in enumerate(cases):
    V = oracle.uniform_rows(1000 + c, 0, n, d); Q = oracle.uniform_rows(5000 + c, 0, nq, d)
    idx = zv.Index(flags=zv.FLAG_BF16_MIRROR); idx.add(V)
    ri, rd, rc = oracle.bruteforce_topk(V, Q, k)
    for mode, name in ((zv.BF_F32, "f32"), (zv.BF_BF16, "bf16"), (zv.BF_I8, "i8")):
        ids, dist, cnt = idx.search_bruteforce(Q, k, mode)
        miss = sum(1 for q in range(nq) if not (np.array_equal(ids[q], ri[q]) and np.array_equal(dist[q], rd[q])))
        print(f"n={n} d={d} k={k} {name}: {miss} of {nq} queries differ", flush=True)
