"""Manual probe of the bf16 register-resident-query stream kernel: compares it with the fp32 route (oracle-parity-tested)
on large shards of several row widths, metrics and k; prints timings.  usage: python tests/manual/bf16_stream_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
import zvdb_amd as zv

cases = [(600_000, 96, 300, 10, "l2"), (600_000, 200, 261, 48, "l2"), (530_003, 500, 300, 10, "l2"), (300_017, 768, 1024, 10, "l2"),
         (300_017, 768, 1000, 100, "l2"), (600_000, 128, 300, 200, "l2"), (600_000, 96, 300, 10, "dot")]
if len(sys.argv) > 1:
    cases = cases[int(sys.argv[1]):int(sys.argv[1]) + 1]
for n, d, nq, k, metric in cases:
    idx = zv.Index(flags=zv.FLAG_BF16_MIRROR, metric=zv.METRIC_DOT if metric == "dot" else zv.METRIC_L2SQ)
    idx.add_synthetic(n, d, 0x5EED)
    Q = oracle.uniform_rows(0xC0FFEE, 0, nq, d)
    ei, ed, ec = idx.search_bruteforce(Q, k, zv.BF_F32)
    t0 = time.time()
    bi, bd, bc = idx.search_bruteforce(Q, k, zv.BF_BF16)
    t1 = time.time()
    bi2, bd2, bc2 = idx.search_bruteforce(Q, k, zv.BF_BF16)
    t2 = time.time()
    same = np.array_equal(ei, bi) and np.array_equal(ed, bd) and np.array_equal(ec, bc)
    rec = np.mean([len(set(bi[i]) & set(ei[i])) / k for i in range(nq)])
    print(f"n={n} d={d} nq={nq} k={k} {metric}: identical to fp32 route: {same}; recall {rec:.4f}; repeat identical: "
          f"{np.array_equal(bi, bi2) and np.array_equal(bd, bd2)}; first call {t1 - t0:.3f}s second {t2 - t1:.3f}s", flush=True)
    if not same:
        bad = [i for i in range(nq) if not np.array_equal(ei[i], bi[i])]
        print("   differing queries:", len(bad), bad[:10])
        q = bad[0]
        print("   f32 :", ei[q][:12], ed[q][:6])
        print("   bf16:", bi[q][:12], bd[q][:6])
    del idx
