# recall of the fp4 screen under the inner-product metrics vs the exact fp32 route (same index)
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, oracle, zvdb_amd as zv
for metric, mname in ((zv.METRIC_L2SQ, "l2"), (zv.METRIC_DOT, "dot"), (zv.METRIC_COSINE, "cos")):
    for d in (128, 768, 1536):
        for n in (30_000, 300_000):
            nq = 64
            V = oracle.uniform_rows(31, 0, n, d) - (0.0 if mname == "l2" else 0.5)
            Q = oracle.uniform_rows(32, 0, nq, d) - (0.0 if mname == "l2" else 0.5)
            idx = zv.Index(metric=metric)
            idx.add(V)
            out = []
            for k in (10, 100):
                ei, ed, _ = idx.search_bruteforce(Q, k, zv.BF_F32)
                ai, ad, _ = idx.search_bruteforce(Q, k, zv.BF_FP4)
                out.append(np.mean([len(set(ai[i]) & set(ei[i])) / k for i in range(nq)]))
            print(f"{mname} d={d} n={n}: fp4 recall@10 {out[0]:.4f} recall@100 {out[1]:.4f}", flush=True)
