"""Pin the CPU oracle against every known answer the reference's own tests hold
for the hot path.  Citations: /root/reference/src/test_hnsw.zig.
The reference is Zig and cannot be run here; these are the hand-derivable
known answers plus the properties its large tests assert."""
import numpy as np
import pytest


def test_distance_is_sequential_fp32_squared_l2(oracle):
    # hnsw.zig:182-192 -- squared, no sqrt; (3,4,5) vs (1,2,3) = 12, vs (4,5,6) = 3
    assert oracle.distance([3, 4, 5], [1, 2, 3]) == np.float32(12)
    assert oracle.distance([3, 4, 5], [4, 5, 6]) == np.float32(3)
    assert oracle.distance([3, 4, 5], [7, 8, 9]) == np.float32(48)
    # strict left-to-right fp32 order, mul rounded before add (no FMA): compare with numpy fp32 loop
    rng = np.random.default_rng(0)
    a = rng.random(777, dtype=np.float32)
    b = rng.random(777, dtype=np.float32)
    s = np.float32(0)
    for i in range(a.size):
        diff = np.float32(a[i] - b[i])
        s = np.float32(s + np.float32(diff * diff))
    assert oracle.distance(a, b) == s


def test_distance_other_types(oracle):
    # test_hnsw.zig:239-273 uses HNSW(i32) and HNSW(f64) (generic T stays on the CPU; out of GPU scope)
    assert oracle.distance_i32([3, 4, 5], [1, 2, 3]) == 12
    assert oracle.distance_f64([3.3, 4.4, 5.5], [1.1, 2.2, 3.3]) == pytest.approx(14.52)


def test_basic_functionality_known_answer(oracle):
    # test_hnsw.zig:24-41; hand-derived: layer-0 adjacency 0:[1] 1:[0,2] 2:[1]; ids [1,0]; dist^2 [3,12]
    for levels in ([0, 0, 0], [3, 0, 2], [-1, -1, -1]):
        h = oracle.HNSW(16, 200)
        for p, lv in zip(([1, 2, 3], [4, 5, 6], [7, 8, 9]), levels):
            h.insert(p, lv)
        assert list(h.neighbors(0, 0)) == [1]
        assert list(h.neighbors(1, 0)) == [0, 2]
        assert list(h.neighbors(2, 0)) == [1]
        ids, dist = h.search([3, 4, 5], 2)
        assert list(ids) == [1, 0]
        assert list(dist) == [3.0, 12.0]


def test_empty_index(oracle):
    # test_hnsw.zig:43-53
    h = oracle.HNSW(16, 200)
    ids, dist = h.search([1, 2, 3], 5)
    assert len(ids) == 0 and len(dist) == 0


def test_single_point(oracle):
    # test_hnsw.zig:55-68
    h = oracle.HNSW(16, 200)
    h.insert([1, 2, 3])
    ids, dist = h.search([1, 2, 3], 1)
    assert list(ids) == [0] and list(dist) == [0.0]
    assert np.array_equal(h.point(0), np.array([1, 2, 3], np.float32))


def test_edge_cases_duplicates_and_k_gt_n(oracle):
    # test_hnsw.zig:104-126: duplicates both returned; k > N returns N results
    h = oracle.HNSW(16, 200)
    h.insert([1, 2, 3])
    h.insert([1, 2, 3])
    ids, dist = h.search([1, 2, 3], 2)
    assert sorted(ids) == [0, 1] and list(dist) == [0.0, 0.0]
    ids, dist = h.search([1, 2, 3], 100)
    assert len(ids) == 2


def test_large_dataset_properties(oracle):
    # test_hnsw.zig:70-102: 10k x 128, k=10 -> len 10, non-decreasing distances
    V = oracle.uniform_rows(0x5EED, 0, 10000, 128)
    h = oracle.HNSW(16, 200)
    for r in V:
        h.insert(r)
    assert h.count() == 10000          # dense ids (test_hnsw.zig:198 analogue)
    q = oracle.uniform_rows(0xC0FFEE, 0, 1, 128)[0]
    ids, dist = h.search(q, 10)
    assert len(ids) == 10
    assert all(dist[i] <= dist[i + 1] for i in range(9))
    assert len(set(ids.tolist())) == 10
    for i, d in zip(ids, dist):
        assert oracle.distance(q, V[int(i)]) == d
    # Consistency, test_hnsw.zig:275-317: same query x10 -> identical results
    for _ in range(9):
        ids2, dist2 = h.search(q, 10)
        assert np.array_equal(ids, ids2) and np.array_equal(dist, dist2)
    # every list obeys the cap m (hnsw.zig:146) and search only used layer 0
    assert max(len(h.neighbors(i, 0)) for i in range(h.count())) <= 16


def test_levels_do_not_affect_search_results(oracle):
    # layer-0 graph and every search result are a pure function of
    # (insertion order, vectors, m, query, k): independent of randomLevel (hnsw.zig:172-180)
    V = oracle.uniform_rows(7, 0, 600, 24)
    Q = oracle.uniform_rows(8, 0, 20, 24)
    hs = [oracle.HNSW(8, 200, level_seed=s) for s in (1, 2, 3)]
    for h in hs:
        for r in V:
            h.insert(r)
    h0 = oracle.HNSW(8, 200)
    for r in V:
        h0.insert(r, 0)
    for q in Q:
        ref = h0.search(q, 7)
        for h in hs:
            got = h.search(q, 7)
            assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
    for i in range(600):
        for h in hs:
            assert np.array_equal(h0.neighbors(i, 0), h.neighbors(i, 0))


def test_reference_search_stats_match_survey_scratch_model(oracle):
    # SURVEY 0.3 / 8c scratch evidence: 10k x 128, m=16, k=10 -> ~77 distance evals per
    # query, mean layer-0 degree ~1.6, recall@10 ~0.04 (first-k-popped search).
    V = oracle.uniform_rows(0x5EED, 0, 10000, 128)
    Q = oracle.uniform_rows(0xC0FFEE, 0, 50, 128)
    h = oracle.HNSW(16, 200)
    for r in V:
        h.insert(r)
    deg = np.mean([len(h.neighbors(i, 0)) for i in range(10000)])
    assert 1.0 < deg < 2.5
    h.reset_counters()
    gt, _, _ = oracle.bruteforce_topk(V, Q, 10)
    hits = 0
    for qi, q in enumerate(Q):
        ids, _ = h.search(q, 10)
        hits += len(set(ids.tolist()) & set(gt[qi].tolist()))
    c = h.counters()
    evals = c.search_dist_evals / len(Q)
    assert 30 < evals < 170
    assert hits / (10 * len(Q)) < 0.3


def test_bruteforce_and_ef_mode_spec(oracle):
    V = oracle.uniform_rows(11, 0, 3000, 32)
    Q = oracle.uniform_rows(12, 0, 16, 32)
    ids, dist, cnt = oracle.bruteforce_topk(V, Q, 10)
    for qi in range(len(Q)):
        d = np.array([oracle.distance(Q[qi], v) for v in V])
        order = np.lexsort((np.arange(len(V)), d))[:10]
        assert np.array_equal(order, ids[qi].astype(np.int64))
        assert np.array_equal(d[order].astype(np.float32), dist[qi])
    # ef-mode on a dense (exact kNN, symmetrised) graph reaches high recall; k<=ef
    h = oracle.HNSW(16, 200)
    for r in V:
        h.insert(r, 0)
    nn, _, _ = oracle.bruteforce_topk(V, V, 9)
    for i in range(len(V)):
        h.set_neighbors(i, 0, [int(x) for x in nn[i] if int(x) != i][:8])
    hits = 0
    for qi in range(len(Q)):
        got, gd = h.search_ef(Q[qi], 10, 200)
        assert all(gd[i] <= gd[i + 1] for i in range(len(gd) - 1))
        hits += len(set(got.tolist()) & set(ids[qi].tolist()))
    assert hits / (10 * len(Q)) > 0.5


def test_merge_topk(oracle):
    rng = np.random.default_rng(3)
    G, nq, k = 4, 5, 10
    dist = np.sort(rng.random((G, nq, k), dtype=np.float32), axis=2)
    ids = rng.permutation(G * nq * k).reshape(G, nq, k).astype(np.uint64)
    cnt = np.full((G, nq), k, dtype=np.uint32)
    cnt[1, 2] = 3
    oi, od, oc = oracle.merge_topk(ids, dist, cnt, k)
    for q in range(nq):
        allp = [(dist[g, q, j], ids[g, q, j]) for g in range(G) for j in range(cnt[g, q])]
        allp.sort()
        assert [p[1] for p in allp[:k]] == oi[q].tolist()


def test_synthetic_stream_is_uniform_and_reproducible(oracle):
    a = oracle.fill_uniform(0x5EED, 0, 100000)
    assert a.min() >= 0.0 and a.max() < 1.0
    assert abs(a.mean() - 0.5) < 0.01 and abs(a.var() - 1 / 12) < 0.005
    b = oracle.fill_uniform(0x5EED, 5000, 100)
    assert np.array_equal(a[5000:5100], b)


def test_search_many_equals_single_searches_on_any_thread_count(oracle):
    """orc_hnsw_search_many (the query-parallel CPU baseline of bench.py) is the reference-compat search without counters: the
    same ids, distances and result counts as one orc_hnsw_search per query, on 1 thread and on 4."""
    V = oracle.uniform_rows(11, 0, 600, 12)
    h = oracle.HNSW(4, 200)
    for r in V:
        h.insert(r, 0)
    Q = oracle.uniform_rows(12, 0, 37, 12)
    want = [h.search(q, 7) for q in Q]
    for threads in (1, 4):
        ids, dist, n = h.search_many(Q, 7, threads=threads)
        for qi, (ri, rd) in enumerate(want):
            assert int(n[qi]) == len(ri) and np.array_equal(ids[qi, :len(ri)], ri) and np.array_equal(dist[qi, :len(ri)], rd)
