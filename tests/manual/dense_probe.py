import sys
sys.path.insert(0, '/root/repo')
import numpy as np, oracle, zvdb_amd as zv
cases = [(4_000_000, 9, 16, 48), (4_000_000, 12, 16, 48), (8_000_000, 16, 16, 48), (8_000_000, 10, 16, 10), (3_000_000, 24, 16, 48), (2_000_000, 32, 16, 48)]
for c, (n, d, nq, k) in enumerate(cases):
    idx = zv.Index(flags=zv.FLAG_BF16_MIRROR); idx.add_synthetic(n, d, 900 + c)
    V = oracle.uniform_rows(900 + c, 0, n, d); Q = oracle.uniform_rows(5900 + c, 0, nq, d)
    ri, rd, rc = oracle.bruteforce_topk(V, Q, k)
    for mode, name in ((zv.BF_F32, "f32"), (zv.BF_BF16, "bf16"), (zv.BF_I8, "i8")):
        ids, dist, cnt = idx.search_bruteforce(Q, k, mode)
        miss = sum(1 for q in range(nq) if not (np.array_equal(ids[q], ri[q]) and np.array_equal(dist[q], rd[q])))
        print(f"n={n} d={d} k={k} {name}: {miss} of {nq} queries differ", flush=True)
