# recall of the fp4 screen under the inner-product metrics vs the exact fp32 route (same index)
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, oracle, zvdb_amd as zv
for metric, mname in ((zv.METRIC_L2SQ, "l2"), (zv.METRIC_DOT, "dot"), (zv.METRIC_COSINE, "cos"), (zv.METRIC_DOT, "dot+"), (zv.METRIC_COSINE, "cos+")):
    for d in (128, 768, 1536):
        for n in (30_000, 300_000):
            nq = 64
            off = 0.0 if mname in ("l2", "dot+", "cos+") else 0.5   # "+": rows and queries of one sign (U[0,1))
            V = oracle.uniform_rows(31, 0, n, d) - off
            Q = oracle.uniform_rows(32, 0, nq, d) - off
            idx = zv.Index(metric=metric)
            idx.add(V)
            out = []
            for k in (10, 100):
                ei, ed, _ = idx.search_bruteforce(Q, k, zv.BF_F32)
                ai, ad, _ = idx.search_bruteforce(Q, k, zv.BF_FP4)
                out.append(np.mean([len(set(ai[i]) & set(ei[i])) / k for i in range(nq)]))
            print(f"{mname} d={d} n={n}: fp4 recall@10 {out[0]:.4f} recall@100 {out[1]:.4f}", flush=True)
