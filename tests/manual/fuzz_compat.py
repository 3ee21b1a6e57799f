# one-off randomized parity sweep of the reference-compat path (insert on every layer + search) against the oracle; the bulk insert
# alternates between its three forms (speculative rounds in one cooperative launch / round by round / one wave in order)
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle, zvdb_amd as zv
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    n = int(rng.choice([1, 2, 3, 17, 100, 500, 2000, 9000]))
    d = int(rng.choice([1, 2, 3, 8, 31, 64, 128, 300]))
    m = int(rng.choice([1, 2, 3, 4, 8, 16, 32]))
    ties = bool(rng.random() < 0.3)
    V = rng.random((n, d), dtype=np.float32)
    if ties: V = np.floor(V * 3).astype(np.float32)
    ref = oracle.HNSW(m, 200, level_seed=int(rng.integers(1, 1000)))
    for r in V: ref.insert(r)
    lv = np.array([ref.node_level(i) for i in range(n)], dtype=np.int32)
    got = zv.HNSW(m, 200)
    split = int(rng.integers(0, n + 1))
    for i in range(min(split, 50)): got.insert(V[i], _level=int(lv[i]))
    s0 = min(split, 50)
    form = ("", "rounds", "0")[c % 3]  # ZVDB_INSERT_SPEC: default = cooperative round loop
    if form: zv.set_tuning("ZVDB_INSERT_SPEC", form)
    else: zv.set_tuning("ZVDB_INSERT_SPEC", None)
    if s0 < n: got.insert_many(V[s0:], lv[s0:])
    zv.set_tuning("ZVDB_INSERT_SPEC", None)
    ok = got.count() == n and got.max_level == ref.max_level
    for i in range(n):
        for layer in range(ref.node_level(i) + 1):
            if not np.array_equal(ref.neighbors(i, layer), got.connections(i, layer)): ok = False
        if not np.array_equal(got.index.get_neighbors(i), ref.neighbors(i, 0)): ok = False
    Q = V[rng.integers(0, n, 6)] + (np.float32(0.0) if ties else np.float32(0.01))
    for q in Q:
        for k in (1, int(rng.integers(1, 40)), 100):
            ri, rd = ref.search(q, k); gi, gd = got.search_ids(q, k)
            if not (np.array_equal(ri, gi) and np.array_equal(rd, gd)): ok = False
    # batched search, counters
    ids, dist, cnt, (ev, pp) = got.index.search_graph(Q, 10, with_stats=True)
    ref.reset_counters()
    for qi, q in enumerate(Q):
        ri, rd = ref.search(q, 10)
        if not (np.array_equal(ids[qi, :len(ri)], ri) and np.array_equal(dist[qi, :len(ri)], rd) and cnt[qi] == len(ri)): ok = False
    cs = ref.counters()
    if (ev, pp) != (cs.search_dist_evals, cs.search_pops): ok = False
    if not ok:
        bad += 1
        print(f"MISMATCH case {c}: n={n} d={d} m={m} ties={ties}", flush=True)
    if c % 10 == 9: print(f"{c + 1} cases, {bad} mismatches", flush=True)
print("done:", cases, "cases,", bad, "mismatches")
