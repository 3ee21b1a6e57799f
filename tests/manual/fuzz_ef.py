# one-off randomized parity sweep of the ef-mode beam search (extension; specification = the oracle's search_ef) on random
# kNN graphs: ids, distances, counts and evaluation / pop counters bit for bit, on fp32 and on bf16 rows, through BOTH beam
# kernels (registers for ef <= 512, LDS beyond or with ZVDB_EF_BEAM=lds); both metrics of the spec (squared L2 and inner product,
# orc_hnsw_search_ef_ex's `metric`), k up to 100, rows up to 1536 dimensions (BASELINE configs[4]: 1536-d, k = 100, ef = 256).
# Search widths 1 .. 4 (ZVDB_OPT_EF_SEARCH_WIDTH) against orc_hnsw_search_ef_w: the side-by-side window of the register-beam kernel
# (width x stride <= 64) and the list-by-list form (wider strides, the LDS-beam kernel, the retry / net passes).
# usage: python tests/manual/fuzz_ef.py [cases] [seed]
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle, zvdb_amd as zv
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    n = int(rng.choice([50, 300, 2000, 6000]))
    d = int(rng.choice([3, 16, 48, 64, 100, 128, 200, 256, 320, 512, 768, 800, 1024, 1536]))
    if d >= 512: n = min(n, 2000)
    dot = bool(rng.random() < 0.4)  # inner-product metric (rows and queries centred on 0)
    m = int(rng.choice([2, 4, 8, 12, 16, 32]))
    m = min(m, n - 1)
    ties = bool(rng.random() < 0.25)
    hb = bool(rng.random() < 0.5)  # bf16 row storage
    V = rng.random((n, d), dtype=np.float32) - np.float32(0.5 if dot else 0.0)
    if ties: V = np.floor(V * 4).astype(np.float32)
    if hb: V = oracle.bf16_round(V)
    om = 1 if dot else 0
    nn, _, _ = oracle.bruteforce_topk(V, V, m + 1, om)
    ref = oracle.HNSW(m, 200)
    for r in V: ref.insert(r, 0)
    adj = np.zeros((n, m), dtype=np.uint32); deg = np.zeros(n, dtype=np.uint32)
    for i in range(n):
        nb = [int(x) for x in nn[i] if int(x) != i][:m]
        ref.set_neighbors(i, 0, nb)
        adj[i, :len(nb)] = nb; deg[i] = len(nb)
    idx = zv.Index(m=m, metric=zv.METRIC_DOT if dot else zv.METRIC_L2SQ, flags=zv.FLAG_BF16_ROWS if hb else 0)
    idx.add(V)
    idx.upload_graph(adj, deg)
    Q = rng.random((16, d), dtype=np.float32) - np.float32(0.5 if dot else 0.0)
    if ties: Q = np.floor(Q * 4).astype(np.float32)
    ok = True
    for _ in range(3):
        k = int(rng.choice([1, 5, 10, 25, 60, 100]))
        width = int(rng.choice([1, 1, 2, 3, 4]))  # search width (ZVDB_OPT_EF_SEARCH_WIDTH; spec: orc_hnsw_search_ef_w)
        idx.set_option(zv.OPT_EF_SEARCH_WIDTH, width)
        ef = int(max(k, rng.choice([1, 10, 33, 64, 100, 128, 129, 200, 256, 300, 512, 600, 1000])))
        try:
            ids, dist, cnt, (ev, pp) = idx.search_graph(Q, k, ef=ef, with_stats=True)
            zv.set_tuning("ZVDB_EF_BEAM", "lds")
            ids2, dist2, cnt2, (ev2, pp2) = idx.search_graph(Q, k, ef=ef, with_stats=True)
            # the visited set sized in two passes: from the fills the calls above recorded, and forced tiny (2^8 slots: most
            # queries stop in the tight pass and are served by the safe retry launch) -- both beam kernels
            # ... and the safe pass with the visited sets in global memory (gv = "1": after a tight LDS pass, and on its own)
            for beam, tp, wpq, gv in (("reg", "", "1", ""), ("lds", "t8", "1", ""), ("reg", "t8", "2", ""), ("reg", "t9", "4", ""), ("reg", "", "2", ""),
                                      ("reg", "", "4", ""), ("reg", "t8", "2", "1"), ("reg", "0", "4", "1"), ("lds", "t9", "1", "1"), ("reg", "0", "2", "net"), ("lds", "t8", "1", "net")):
                zv.set_tuning("ZVDB_EF_BEAM", beam); zv.set_tuning("ZVDB_EF_TWO_PASS", tp); zv.set_tuning("ZVDB_EF_WPQ", wpq)  # waves per query
                if gv: zv.set_tuning("ZVDB_EF_GLOBAL_VISITED", gv)
                else: zv.set_tuning("ZVDB_EF_GLOBAL_VISITED", None)
                # (gv = "net": the SAFE LDS table forced to 2^8 slots too -- the queries that fill it go through the net pass,
                # visited set in a global table handed out on demand)
                if gv == "net": zv.set_tuning("ZVDB_EF_GLOBAL_VISITED", None); zv.set_tuning("ZVDB_EF_HASH_BITS", "8")
                else: zv.set_tuning("ZVDB_EF_HASH_BITS", None)
                # large beams (ef > 256): the packed visited table off / with more (small) regions than the ids need; 8 waves per query
                pk = [None, "0", "r3", "r6"][int(rng.integers(0, 4))] if ef > 256 else None
                zv.set_tuning("ZVDB_EF_PACKED_VISITED", "0" if pk == "0" else None); zv.set_tuning("ZVDB_EF_PACKED_R", pk[1:] if pk and pk[0] == "r" else None)
                if ef > 256 and width > 1 and wpq == "4" and rng.random() < 0.5: zv.set_tuning("ZVDB_EF_WPQ", "8")
                ids3, dist3, cnt3, (ev3, pp3) = idx.search_graph(Q, k, ef=ef, with_stats=True)
                if not (np.array_equal(ids, ids3) and np.array_equal(dist, dist3) and np.array_equal(cnt, cnt3) and (ev, pp) == (ev3, pp3)):
                    ok = False; print(f"case {c}: two-pass (packed visited {pk!r}) visited sizing differs (beam {beam}, ZVDB_EF_TWO_PASS={tp!r}, ZVDB_EF_WPQ={wpq}, ZVDB_EF_GLOBAL_VISITED={gv!r}) stats {(ev3, pp3)} vs {(ev, pp)}", flush=True)
        except zv.ZvdbError as e:
            print(f"case {c}: n={n} d={d} m={m} k={k} ef={ef}: {e}"); continue
        finally:
            zv.set_tuning("ZVDB_EF_BEAM", None); zv.set_tuning("ZVDB_EF_TWO_PASS", None); zv.set_tuning("ZVDB_EF_WPQ", None); zv.set_tuning("ZVDB_EF_GLOBAL_VISITED", None); zv.set_tuning("ZVDB_EF_HASH_BITS", None)
            zv.set_tuning("ZVDB_EF_PACKED_VISITED", None); zv.set_tuning("ZVDB_EF_PACKED_R", None)
        if not (np.array_equal(ids, ids2) and np.array_equal(dist, dist2) and np.array_equal(cnt, cnt2) and (ev, pp) == (ev2, pp2)):
            ok = False
            print(f"case {c}: the two beam kernels differ: ids {np.array_equal(ids, ids2)} dist {np.array_equal(dist, dist2)} cnt {np.array_equal(cnt, cnt2)} "
                  f"stats reg {(ev, pp)} lds {(ev2, pp2)}", flush=True)
            ref.reset_counters()
            r = [ref.search_ef(q, k, ef, om, 0, width) for q in Q]
            cs = ref.counters()
            print("   oracle stats", (cs.search_dist_evals, cs.search_pops), "reg rows equal:", [bool(np.array_equal(ids[i, :len(r[i][0])], r[i][0])) for i in range(len(Q))],
                  "lds rows equal:", [bool(np.array_equal(ids2[i, :len(r[i][0])], r[i][0])) for i in range(len(Q))], flush=True)
        ref.reset_counters()
        for qi, q in enumerate(Q):
            ri, rd = ref.search_ef(q, k, ef, om, 0, width)
            if not (cnt[qi] == len(ri) and np.array_equal(ids[qi, :len(ri)], ri) and np.array_equal(dist[qi, :len(ri)], rd)): ok = False
        cs = ref.counters()
        if (ev, pp) != (cs.search_dist_evals, cs.search_pops): ok = False
        if not ok: print(f"MISMATCH case {c}: n={n} d={d} m={m} ties={ties} bf16_rows={hb} dot={dot} k={k} ef={ef} width={width}", flush=True); break
    bad += 0 if ok else 1
print("done:", cases, "cases,", bad, "mismatches")
