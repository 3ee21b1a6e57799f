"""CPU tests of the EXTENSIONS' specifications in the oracle (no reference symbol exists for them: parity unpinned by the
reference; the HIP path is tested against these functions on the GPU).  Here the C functions are checked against a
second, deliberately naive pure-Python restatement of the same written rules, and against simple invariants."""
import numpy as np
import pytest


def _d(a, b, metric):
    a = np.asarray(a, dtype=np.float32); b = np.asarray(b, dtype=np.float32)
    s = np.float32(0)
    for x, y in zip(a, b):
        if metric == 0:
            t = np.float32(x - y); s = np.float32(s + np.float32(t * t))
        else:
            s = np.float32(s + np.float32(x * y))
    return s if metric == 0 else np.float32(-s)


def _select(V, cand, cap, metric):
    """cand: [(dist, id)] ascending.  The rule of oracle/zvdb_oracle.c:orc_graph_select, written out naively."""
    kept, aside = [], []
    for dc, c in cand:
        if len(kept) == cap:
            break
        if all(dc < _d(V[c], V[s], metric) for _, s in kept):
            kept.append((dc, c))
        else:
            aside.append((dc, c))
    for e in aside:
        if len(kept) == cap:
            break
        kept.append(e)
    return kept


@pytest.mark.parametrize("n,d,M,efc,metric,grid", [(60, 4, 3, 12, 0, False), (80, 3, 4, 20, 0, True), (50, 5, 2, 49, 1, False), (30, 2, 6, 8, 0, True)])
def test_bulk_layer0_matches_a_naive_restatement(oracle, n, d, M, efc, metric, grid):
    rng = np.random.default_rng(n * 7 + d)
    V = rng.random((n, d), dtype=np.float32) - np.float32(0.5 if metric else 0.0)
    if grid:
        V = np.floor(V * 3).astype(np.float32)
    pool = min(efc, n - 1)
    # candidates: exact nearest by (distance, id), self excluded
    cand = []
    for u in range(n):
        lst = sorted(((_d(V[u], V[v], metric), v) for v in range(n) if v != u))[:pool]
        cand.append(lst)
    ci = np.array([[c for _, c in l] for l in cand], dtype=np.uint64)
    cd = np.array([[x for x, _ in l] for l in cand], dtype=np.float32)
    oi, od = oracle._knn_excluding_self(V, np.arange(n), pool, metric, threads=2)
    assert np.array_equal(oi, ci) and np.array_equal(od, cd)
    M0, cmax = 2 * M, 128
    adj, deg = oracle.graph_build_layer0(V, ci, cd, M, M0, cmax, metric)
    S = [_select(V, cand[u], M, metric) for u in range(n)]
    for v in range(n):
        own = {c for _, c in S[v]}
        C = list(S[v]) + [(dist, u) for u in range(n) for dist, c in S[u] if c == v and u not in own]
        C = sorted(C)[:cmax]
        want = [c for _, c in _select(V, C, M0, metric)]
        assert list(adj[v, :deg[v]]) == want, v


def test_bulk_levels_and_descent_rules(oracle):
    lv = oracle.bulk_levels(123, 20000)
    frac = [(lv >= L).mean() for L in range(1, 6)]
    assert all(abs(f - 0.5 ** L) < 0.02 for L, f in zip(range(1, 6), frac))      # each further level with probability 1/2 (hnsw.zig:172-180)
    assert lv.max() <= 31 and np.array_equal(lv, oracle.bulk_levels(123, 20000))   # deterministic in (seed, id)
    V = oracle.uniform_rows(9, 0, 400, 6)
    h, levels, top = oracle.build_bulk_graph(V, 4, 16, 77)
    assert levels[top] == levels.max() and top == int(np.argmax(levels))
    for i in range(400):   # a layer's list holds nodes of that layer only, never the node itself, at most M (2 M at layer 0)
        for L in range(int(levels[i]) + 1):
            nb = h.neighbors(i, L)
            assert i not in nb and len(nb) <= (8 if L == 0 else 4) and all(levels[int(x)] >= L for x in nb)
    q = oracle.uniform_rows(10, 0, 1, 6)[0]
    e = h.descend(q, 0, top)
    # the descent ends where no layer-1 neighbour of the node is closer (blocked traversal distance), and never gets worse than the top
    lib = oracle.lib()
    import ctypes as C
    f32p = C.POINTER(C.c_float)
    bd = lambda x: lib.orc_distance_blocked16_f32(q.ctypes.data_as(f32p), np.ascontiguousarray(V[x]).ctypes.data_as(f32p), 6)
    assert bd(e) <= bd(top)
    if levels[e] >= 1:
        assert all(bd(int(x)) >= bd(e) for x in h.neighbors(e, 1))
    ids, dist = h.search_ef(q, 5, 32, 0, e)
    gt, gd, _ = oracle.bruteforce_topk(V, q.reshape(1, -1), 5)
    assert len(set(ids.tolist()) & set(gt[0].tolist())) >= 4 and np.all(np.diff(dist) >= 0)


def test_ef_spec_metric_and_entry(oracle):
    """orc_hnsw_search_ef_ex: metric 1 ranks by the negated inner product; the entry node is a parameter; ef = N on a connected
    graph returns the exact top-k under either metric."""
    n, d, m = 300, 8, 6
    V = oracle.uniform_rows(5, 0, n, d) - np.float32(0.5)
    h = oracle.HNSW(m, 200)
    for r in V:
        h.insert(r, 0)
    nn, _, _ = oracle.bruteforce_topk(V, V, m + 1)
    for i in range(n):
        h.set_neighbors(i, 0, [int(x) for x in nn[i] if int(x) != i][:m] + [(i + 1) % n])  # (+ a ring: connected)
    q = oracle.uniform_rows(6, 0, 1, d)[0] - np.float32(0.5)
    for metric in (0, 1):
        gt, gd, _ = oracle.bruteforce_topk(V, q.reshape(1, -1), 10, metric=metric)
        for entry in (0, 17, n - 1):
            ids, dist = h.search_ef(q, 10, n, metric, entry)
            assert np.array_equal(ids, gt[0]) and np.array_equal(dist, gd[0]), (metric, entry)


def _line_graph(oracle):
    """Six points on a line (x = 0 .. 5), layer-0 lists 0:[1,2] 1:[0,3] 2:[0,4] 3:[1,5] 4:[2] 5:[3]."""
    h = oracle.HNSW(2, 200)
    for x in range(6):
        h.insert(np.array([x], dtype=np.float32), 0)
    for i, lst in {0: [1, 2], 1: [0, 3], 2: [0, 4], 3: [1, 5], 4: [2], 5: [3]}.items():
        h.set_neighbors(i, 0, lst)
    return h


def test_search_width_hand_derived(oracle):
    """orc_hnsw_search_ef_w on a graph small enough to run on paper (query x = 4.6, entry 0, ef = 2, k = 2):
    width 1: W = [0]; pop 0 -> 1, 2 evaluated, W = [2, 1]; pop 2 -> 4 in, W = [4, 2] (1 falls off unexpanded); pop 4 -> nothing new;
             stop: 3 pops, 4 evaluations, result [4, 2].
    width 2: pop 0 -> W = [2, 1]; pops 2 AND 1 together (both marked before either list is read): 2's list brings 4, W = [4, 2];
             1's list brings 3 (2.56 < 6.76), W = [4, 3]; pops 4 and 3: 3's list brings 5, W = [5, 4]; pop 5: nothing;
             stop: 6 pops, 6 evaluations, result [5, 4] -- the node the plain search never reaches."""
    h = _line_graph(oracle)
    q = np.array([4.6], dtype=np.float32)
    d = lambda i: oracle.distance(q, np.array([i], dtype=np.float32))
    h.reset_counters()
    ids, dist = h.search_ef(q, 2, 2, 0, 0, 1)
    c = h.counters()
    assert [int(x) for x in ids] == [4, 2] and list(dist) == [d(4), d(2)]
    assert (c.search_pops, c.search_dist_evals) == (3, 4)
    h.reset_counters()
    ids, dist = h.search_ef(q, 2, 2, 0, 0, 2)
    c = h.counters()
    assert [int(x) for x in ids] == [5, 4] and list(dist) == [d(5), d(4)]
    assert (c.search_pops, c.search_dist_evals) == (6, 6)
    # a width beyond the beam's unexpanded entries takes what there is; width 0 means 1
    assert [int(x) for x in h.search_ef(q, 2, 2, 0, 0, 8)[0]] == [5, 4]
    assert [int(x) for x in h.search_ef(q, 2, 2, 0, 0, 0)[0]] == [4, 2]


def test_search_width_one_is_the_plain_beam_search(oracle):
    """width = 1 of orc_hnsw_search_ef_w is orc_hnsw_search_ef_ex (which now calls it), on a random bulk graph, counters included;
    larger widths return beams of the same quality on data with neighbourhoods (recall within a point of width 1)."""
    rng = np.random.default_rng(9)
    V = (rng.random((1500, 3), dtype=np.float32) @ rng.random((3, 24), dtype=np.float32)).astype(np.float32)
    ref, levels, top = oracle.build_bulk_graph(V, 8, 32, 5)
    Q = (rng.random((30, 3), dtype=np.float32) @ rng.random((3, 24), dtype=np.float32)).astype(np.float32)
    gt, _, _ = oracle.bruteforce_topk(V, Q, 10)
    rec = {}
    for w in (1, 2, 4):
        hits = 0
        for qi, q in enumerate(Q):
            ids, _ = ref.search_ef(q, 10, 48, 0, ref.descend(q, 0, top), w)
            hits += len(set(int(x) for x in ids) & set(int(x) for x in gt[qi]))
            if w == 1:
                a = ref._L.orc_hnsw_search_ef_ex  # the un-widthed entry point returns the same list
                import ctypes as C
                oi = np.zeros(10, dtype=np.uint64); od = np.zeros(10, dtype=np.float32)
                n = a(ref._h, q.ctypes.data_as(C.POINTER(C.c_float)), q.size, 10, 48, 0, ref.descend(q, 0, top),
                      oi.ctypes.data_as(C.POINTER(C.c_uint64)), od.ctypes.data_as(C.POINTER(C.c_float)))
                assert np.array_equal(oi[:n], ids)
        rec[w] = hits / (len(Q) * 10)
    assert rec[1] >= 0.9 and abs(rec[2] - rec[1]) <= 0.02 and abs(rec[4] - rec[1]) <= 0.02, rec
