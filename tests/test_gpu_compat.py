"""GPU parity tests of the reference-compat path (a1-a8): the HIP kernels behind the C ABI vs the
CPU oracle on the same seeded inputs -- bit-exact ids AND distances.  The first block restates the
reference's own tests (src/test_hnsw.zig) against the drop-in `HNSW` mirror."""
import threading

import numpy as np
import pytest
from conftest import tuning_prelude

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zv():
    import zvdb_amd
    assert zvdb_amd.device_count() > 0, "no HIP device: the product path has no CPU fallback"
    return zvdb_amd


def euclid(a, b):
    return float(np.sqrt(np.sum((np.asarray(a, np.float32) - np.asarray(b, np.float32)) ** 2)))


# ---- the reference's tests, restated (src/test_hnsw.zig) ------------------------------------
def test_basic_functionality(zv):  # test_hnsw.zig:24-41
    h = zv.HNSW(16, 200)
    for p in ([1, 2, 3], [4, 5, 6], [7, 8, 9]):
        h.insert(p)
    q = [3, 4, 5]
    res = h.search(q, 2)
    assert len(res) == 2
    assert euclid(q, res[0].point) <= euclid(q, res[1].point)
    assert [r.id for r in res] == [1, 0] and [r.distance for r in res] == [3.0, 12.0]


def test_empty_index(zv):  # test_hnsw.zig:43-53
    h = zv.HNSW(16, 200)
    assert h.search([1, 2, 3], 5) == []


def test_single_point(zv):  # test_hnsw.zig:55-68
    h = zv.HNSW(16, 200)
    h.insert([1, 2, 3])
    res = h.search([1, 2, 3], 1)
    assert len(res) == 1 and np.array_equal(res[0].point, np.array([1, 2, 3], np.float32))


def test_edge_cases(zv):  # test_hnsw.zig:104-126
    h = zv.HNSW(16, 200)
    h.insert([1, 2, 3])
    h.insert([1, 2, 3])
    res = h.search([1, 2, 3], 2)
    assert len(res) == 2
    assert all(np.array_equal(r.point, np.array([1, 2, 3], np.float32)) for r in res)
    assert len(h.search([1, 2, 3], 100)) == 2


def test_dimension_mismatch_panics(zv):  # hnsw.zig:183-185
    h = zv.HNSW(16, 200)
    h.insert([1, 2, 3])
    with pytest.raises(zv.ZvdbPanic):
        h.insert([1, 2, 3, 4])
    with pytest.raises(zv.ZvdbPanic):
        h.search([1, 2], 1)


def test_large_dataset_and_consistency(zv, oracle):  # test_hnsw.zig:70-102, :275-317
    V = oracle.uniform_rows(0x5EED, 0, 3000, 128)
    h = zv.HNSW(16, 200)
    for r in V:
        h.insert(r)
    assert h.count() == 3000
    q = oracle.uniform_rows(0xC0FFEE, 0, 1, 128)[0]
    res = h.search(q, 10)
    assert len(res) == 10
    last = 0.0
    for r in res:
        d = euclid(q, r.point)
        assert d >= last
        last = d
    first = [r.point.copy() for r in res]
    for _ in range(9):
        again = h.search(q, 10)
        assert all(np.array_equal(a.point, b) for a, b in zip(again, first))


def test_concurrent_access(zv):  # test_hnsw.zig:154-209 (8 threads; fewer points per thread)
    h = zv.HNSW(16, 200)
    rng = np.random.default_rng(1)
    data = rng.random((8, 100, 128), dtype=np.float32)

    def work(t):
        for p in data[t]:
            h.insert(p)

    ts = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert h.count() == 800
    assert len(h.search(rng.random(128, dtype=np.float32), 10)) == 10


# ---- parity with the oracle -------------------------------------------------------------------
@pytest.mark.parametrize("n,d,m", [(600, 24, 4), (1500, 128, 16), (400, 3, 2), (300, 70, 16)])
def test_insert_and_search_bit_exact_vs_oracle(zv, oracle, n, d, m):
    """Full insert path (a2/a3/a4) + search (a5/a6): identical graphs on every layer and identical
    (ids, distances, E, P) for every query and k."""
    V = oracle.uniform_rows(100 + n, 0, n, d)
    Q = oracle.uniform_rows(200 + n, 0, 24, d)
    ref = oracle.HNSW(m, 200, level_seed=7)
    got = zv.HNSW(m, 200)
    for r in V:
        ref.insert(r)  # oracle draws the level; feed the same level to the product
        got.insert(r, _level=ref.node_level(ref.count() - 1))
    assert got.count() == ref.count() and got.max_level == ref.max_level
    for i in range(n):
        for layer in range(ref.node_level(i) + 1):
            assert np.array_equal(ref.neighbors(i, layer), got.connections(i, layer)), (i, layer)
        assert got.connections(i, ref.node_level(i) + 1) is None
        assert np.array_equal(got.index.get_neighbors(i), ref.neighbors(i, 0))  # device mirror
    for k in (1, 10, 25, 100):
        for q in Q:
            ref.reset_counters()
            ri, rd = ref.search(q, k)
            gi, gd = got.search_ids(q, k)
            assert np.array_equal(ri, gi) and np.array_equal(rd, gd), k
            c = ref.counters()
            assert (got.last_evals, got.last_pops) == (c.search_dist_evals, c.search_pops)


def test_distances_seam_bit_exact(zv, oracle):
    V = oracle.uniform_rows(5, 0, 500, 77)
    idx = zv.Index()
    idx.add(V)
    q = oracle.uniform_rows(6, 0, 1, 77)[0]
    ids = np.array([0, 499, 17, 17, 250] + list(range(100, 200)), dtype=np.uint64)
    want = np.array([oracle.distance(q, V[int(i)]) for i in ids], dtype=np.float32)
    assert np.array_equal(idx.distances(q, ids), want)
    want2 = np.array([oracle.distance(V[3], V[int(i)]) for i in ids], dtype=np.float32)
    assert np.array_equal(idx.distances_from(3, ids), want2)
    with pytest.raises(zv.ZvdbError):
        idx.distances(q, [500])            # error.NodeNotFound
    with pytest.raises(zv.ZvdbPanic):
        idx.distances(q[:5], [1])          # dimension mismatch panics


def _graph_index(zv, oracle, n, d, m, seed):
    V = oracle.uniform_rows(seed, 0, n, d)
    ref = oracle.HNSW(m, 200)
    for r in V:
        ref.insert(r, 0)
    idx = zv.Index(m=m)
    idx.add(V)
    adj, deg = ref.layer0(m)
    idx.upload_graph(adj, deg)
    return V, ref, idx


def test_batched_compat_search_bit_exact(zv, oracle):
    V, ref, idx = _graph_index(zv, oracle, 10000, 128, 16, 0x5EED)  # BASELINE config 1 shape
    Q = oracle.uniform_rows(0xC0FFEE, 0, 256, 128)
    for k in (10, 50):
        ids, dist, cnt, (ev, pp) = idx.search_graph(Q, k, with_stats=True)
        ref.reset_counters()
        for qi, q in enumerate(Q):
            ri, rd = ref.search(q, k)
            assert cnt[qi] == len(ri)
            assert np.array_equal(ids[qi, :len(ri)], ri) and np.array_equal(dist[qi, :len(ri)], rd)
        c = ref.counters()
        assert (ev, pp) == (c.search_dist_evals, c.search_pops)


def test_exact_ties_follow_oracle_heap_order(zv, oracle):
    """Coordinates on a coarse integer grid => many EXACTLY equal fp32 distances; the pop order then
    depends on the restated Zig std.PriorityQueue (parity unpinned by the reference, pinned to the
    oracle here)."""
    rng = np.random.default_rng(9)
    V = rng.integers(0, 3, size=(1500, 6)).astype(np.float32)
    ref = oracle.HNSW(6, 200)
    for r in V:
        ref.insert(r, 0)
    idx = zv.Index(m=6)
    idx.add(V)
    adj, deg = ref.layer0(6)
    idx.upload_graph(adj, deg)
    Q = rng.integers(0, 3, size=(64, 6)).astype(np.float32)
    ids, dist, cnt = idx.search_graph(Q, 20)
    for qi, q in enumerate(Q):
        ri, rd = ref.search(q, 20)
        assert np.array_equal(ids[qi, :cnt[qi]], ri) and np.array_equal(dist[qi, :cnt[qi]], rd)


def test_ef_mode_bit_exact_vs_oracle_spec(zv, oracle, tuning):
    n, d, m = 4000, 48, 12
    V = oracle.uniform_rows(31, 0, n, d)
    nn, _, _ = oracle.bruteforce_topk(V, V, m + 1)
    ref = oracle.HNSW(m, 200)
    for r in V:
        ref.insert(r, 0)
    adj = np.zeros((n, m), dtype=np.uint32)
    deg = np.zeros(n, dtype=np.uint32)
    for i in range(n):
        nb = [int(x) for x in nn[i] if int(x) != i][:m]
        ref.set_neighbors(i, 0, nb)
        adj[i, :len(nb)] = nb
        deg[i] = len(nb)
    idx = zv.Index(m=m)
    idx.add(V)
    idx.upload_graph(adj, deg)
    Q = oracle.uniform_rows(32, 0, 64, d)
    for k, ef in ((10, 10), (10, 64), (25, 200)):
        # 1st call: the safe visited table (no fill seen yet for this ef); 2nd: the tight table sized IN SLOTS from the fill the
        # first call left (ef_search.hip: launch_search_ef); 3rd / 4th: tight tables forced so small that most queries fill
        # them and are served again by the safe pass -- a power-of-two one and one of 300 slots.  Same results, same counters.
        for env in (None, None, "t8", "s300"):
            tuning("ZVDB_EF_TWO_PASS", env)
            ids, dist, cnt, (ev, pp) = idx.search_graph(Q, k, ef=ef, with_stats=True)
            ref.reset_counters()
            for qi, q in enumerate(Q):
                ri, rd = ref.search_ef(q, k, ef)
                assert cnt[qi] == len(ri)
                assert np.array_equal(ids[qi, :len(ri)], ri) and np.array_equal(dist[qi, :len(ri)], rd)
            c = ref.counters()
            assert (ev, pp) == (c.search_dist_evals, c.search_pops), env
        tuning("ZVDB_EF_TWO_PASS", None)


def test_greedy_layer0_seam(zv, oracle):
    V, ref, idx = _graph_index(zv, oracle, 2000, 16, 8, 77)
    # descent for an existing row from the entry point == what insert would do at layer 0
    for owner in (5, 1234, 1999):
        ep, dist = 0, oracle.distance(V[owner], V[0])
        changed = True
        evals = 0
        while changed:
            changed = False
            for nb in ref.neighbors(ep, 0):
                evals += 1
                dd = oracle.distance(V[owner], V[int(nb)])
                if dd < dist:
                    cand, dist, changed = int(nb), dd, True
            if changed:
                ep = cand
        # (the python loop above moves ep only after the full scan, like hnsw.zig:92-101)
        g_ep, g_dist, g_ev = idx.greedy_layer0(owner, 0, float(oracle.distance(V[owner], V[0])))
        assert (g_ep, np.float32(g_dist)) == (ep, np.float32(dist))


def test_gpu_knn_graph_build_and_ef_search(zv, oracle):
    """batched GPU construction of the layer-0 graph (exact kNN lists via the
    brute-force MFMA path) + ef-mode search over it; both pinned to the oracle (exact kNN by
    (distance, id); ef-mode spec)."""
    n, d, M = 3000, 32, 8
    V = oracle.uniform_rows(91, 0, n, d)
    idx = zv.Index(m=M, flags=zv.FLAG_BF16_MIRROR)
    idx.add(V)
    for mode in (zv.BF_F32, zv.BF_I8):
        idx.build_knn_graph(M, mode)
        nn, _, _ = oracle.bruteforce_topk(V, V, M + 1)
        ref = oracle.HNSW(M, 200)
        for r in V:
            ref.insert(r, 0)
        for i in range(n):
            want = [int(x) for x in nn[i] if int(x) != i][:M]
            assert [int(x) for x in idx.get_neighbors(i)] == want, i
            ref.set_neighbors(i, 0, want)
    Q = oracle.uniform_rows(92, 0, 48, d)
    ids, dist, cnt = idx.search_graph(Q, 10, ef=64)
    gt, _, _ = oracle.bruteforce_topk(V, Q, 10)
    hits = 0
    for qi, q in enumerate(Q):
        ri, rd = ref.search_ef(q, 10, 64)
        assert np.array_equal(ids[qi, :len(ri)], ri) and np.array_equal(dist[qi, :len(ri)], rd)
        hits += len(set(ri.tolist()) & set(gt[qi].tolist()))
    assert hits / (10 * len(Q)) > 0.6


@pytest.mark.parametrize("n,d,m", [(500, 2, 3), (400, 3, 1), (300, 4, 16), (64, 1, 2)])
def test_fused_insert_under_exact_ties(zv, oracle, n, d, m):
    """zvdb_hip_insert_layer0 (a2 + a4 + a3 in one launch) on data made of ties: points on a 3-wide integer grid with many
    duplicates, tiny m (every connect shrinks).  Strict '<' in the scan (earliest wins), stable shrink order: the graphs must
    equal the oracle's on every layer, and so must every search."""
    rng = np.random.default_rng(n)
    V = np.floor(rng.random((n, d), dtype=np.float32) * 3).astype(np.float32)
    ref = oracle.HNSW(m, 200, level_seed=3)
    got = zv.HNSW(m, 200)
    for i, r in enumerate(V):
        ref.insert(r)
        got.insert(r, _level=ref.node_level(i))
    assert got.max_level == ref.max_level
    for i in range(n):
        for layer in range(ref.node_level(i) + 1):
            assert np.array_equal(ref.neighbors(i, layer), got.connections(i, layer)), (i, layer)
        assert np.array_equal(got.index.get_neighbors(i), ref.neighbors(i, 0))
    for q in V[:20] + np.float32(0.25):
        for k in (1, 7, 40):
            ri, rd = ref.search(q, k)
            gi, gd = got.search_ids(q, k)
            assert np.array_equal(ri, gi) and np.array_equal(rd, gd)


@pytest.mark.parametrize("form", ["", "rounds", "0"])
@pytest.mark.parametrize("n,d,m,ties", [(3000, 32, 16, False), (700, 2, 3, True), (9000, 8, 4, False)])
def test_insert_many_equals_sequential_inserts(zv, oracle, tuning, n, d, m, ties, form):
    """HNSW.insert_many (zvdb_hip_insert_batch: a chunk of inserts, each seeing the lists its predecessors left) builds the
    graph of n insert() calls: identical to the oracle's on every layer, device mirror included; also when the index already
    holds points and when a chunk boundary (8192) falls inside the batch.  In all three forms of the bulk insert
    (ZVDB_INSERT_SPEC): speculative parallel rounds in one cooperative launch (default), round by round from the host, and
    one wave in order -- on tie-heavy data with tiny m too, where nearly every point of a round ends on the same few nodes."""
    tuning("ZVDB_INSERT_SPEC", form or None)
    rng = np.random.default_rng(n)
    V = rng.random((n, d), dtype=np.float32)
    if ties:
        V = np.floor(V * 3).astype(np.float32)
    ref = oracle.HNSW(m, 200, level_seed=11)
    for r in V:
        ref.insert(r)
    levels = np.array([ref.node_level(i) for i in range(n)], dtype=np.int32)
    got = zv.HNSW(m, 200)
    head = 5
    for i in range(head):  # a few single inserts first, then the rest in two batches
        got.insert(V[i], _level=int(levels[i]))
    got.insert_many(V[head:n // 2], levels[head:n // 2])
    got.insert_many(V[n // 2:], levels[n // 2:])
    assert got.count() == n and got.max_level == ref.max_level
    for i in range(n):
        for layer in range(ref.node_level(i) + 1):
            assert np.array_equal(ref.neighbors(i, layer), got.connections(i, layer)), (i, layer)
        if i % 7 == 0:
            assert np.array_equal(got.index.get_neighbors(i), ref.neighbors(i, 0))
    fresh = zv.HNSW(m, 200)  # an empty index: the first node goes through the single-insert path
    fresh.insert_many(V[:300], levels[:300])
    ref2 = oracle.HNSW(m, 200, level_seed=11)
    for r in V[:300]:
        ref2.insert(r)
    for i in range(300):
        assert np.array_equal(ref2.neighbors(i, 0), fresh.connections(i, 0))
    for q in V[:10] + np.float32(0.125):
        ri, rd = ref.search(q, 10)
        gi, gd = got.search_ids(q, 10)
        assert np.array_equal(ri, gi) and np.array_equal(rd, gd)


def test_step_by_step_binding_still_bit_exact(oracle):
    """ZVDB_HOST_INSERT=steps: the host mirror drives the insert the way INTEGRATION.md's Zig binding does (layer 0 fused on
    the device through zvdb_hip_insert_layer0 / _batch_layer0, upper layers a host loop over zvdb_hip_distances_from) --
    the same graphs as the whole-insert-on-device default, i.e. the oracle's."""
    import os, subprocess, sys
    code = (
        "import numpy as np, oracle, zvdb_amd as zv\n"
        "n, d, m = 1200, 16, 4\n"
        "V = oracle.uniform_rows(5, 0, n, d)\n"
        "ref = oracle.HNSW(m, 200, level_seed=9)\n"
        "for r in V: ref.insert(r)\n"
        "lv = np.array([ref.node_level(i) for i in range(n)], dtype=np.int32)\n"
        "a = zv.HNSW(m, 200)\n"
        "for i in range(400): a.insert(V[i], _level=int(lv[i]))\n"
        "a.insert_many(V[400:], lv[400:])\n"
        "for i in range(n):\n"
        "    for layer in range(ref.node_level(i) + 1):\n"
        "        assert np.array_equal(ref.neighbors(i, layer), a.connections(i, layer)), (i, layer)\n"
        "    assert np.array_equal(a.index.get_neighbors(i), ref.neighbors(i, 0))\n"
        "for q in V[:8] + np.float32(0.5):\n"
        "    ri, rd = ref.search(q, 10); gi, gd = a.search_ids(q, 10)\n"
        "    assert np.array_equal(ri, gi) and np.array_equal(rd, gd)\n"
        "print('steps ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", tuning_prelude({"ZVDB_HOST_INSERT": "steps"}) + code], cwd=root,
                         capture_output=True, text=True)
    assert out.returncode == 0 and "steps ok" in out.stdout, out.stdout + out.stderr


def test_heap_emulation_alone_equals_the_two_kernel_search(zv, oracle):
    """ZVDB_COMPAT_SEARCH=heap (every query on the PriorityQueue emulation) returns what the default (unordered-frontier
    kernel + gated emulation for equal minima) returns, on distinct AND on tie-heavy data, counters included."""
    import os, subprocess, sys
    code = (
        "import numpy as np, oracle, zvdb_amd as zv\n"
        "rng = np.random.default_rng(3)\n"
        "out = []\n"
        "for ties in (False, True):\n"
        "    V = rng.random((3000, 6), dtype=np.float32)\n"
        "    if ties: V = np.floor(V * 3).astype(np.float32)\n"
        "    h = zv.HNSW(8, 200)\n"
        "    h.insert_many(V, np.zeros(len(V), dtype=np.int32))\n"
        "    Q = V[:200] + np.float32(0.0 if ties else 0.01)\n"
        "    ids, dist, cnt, (ev, pp) = h.index.search_graph(Q, 12, with_stats=True)\n"
        "    out.append((ids, dist, cnt, ev, pp))\n"
        "np.save('/tmp/zvdb_compat_' + ('heap' if zv._lib.hip().zvdb_hip_get_tuning(b'ZVDB_COMPAT_SEARCH') else 'fast') + '.npy',\n"
        "        np.array([np.concatenate([o[0].ravel().astype(np.float64), o[1].ravel().astype(np.float64), o[2].ravel().astype(np.float64), [o[3], o[4]]]) for o in out]))\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env in ({}, {"ZVDB_COMPAT_SEARCH": "heap"}):
        r = subprocess.run([sys.executable, "-c", tuning_prelude(env) + code], cwd=root, capture_output=True, text=True)
        assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    a, b = np.load("/tmp/zvdb_compat_fast.npy"), np.load("/tmp/zvdb_compat_heap.npy")
    assert np.array_equal(a, b)


def test_hnsw_with_bf16_valued_rows_uses_the_step_binding(zv, oracle):
    """An index whose rows are rounded to bf16 on insert (ZVDB_FLAG_BF16_ROWS) is outside zvdb_hip_insert_batch's case; the
    host mirror then drives the insert step by step.  The graph equals the oracle's built from the rounded points."""
    n, d, m = 500, 20, 4
    V = oracle.uniform_rows(77, 0, n, d)
    R = oracle.bf16_round(V)
    ref = oracle.HNSW(m, 200, level_seed=5)
    got = zv.HNSW(m, 200, flags=zv.FLAG_BF16_ROWS)
    for i in range(n):
        ref.insert(R[i])
        got.insert(V[i], _level=ref.node_level(i))
    for i in range(n):
        for layer in range(ref.node_level(i) + 1):
            assert np.array_equal(ref.neighbors(i, layer), got.connections(i, layer)), (i, layer)
    q = V[3] + np.float32(0.05)
    ri, rd = ref.search(q, 10)
    gi, gd = got.search_ids(q, 10)
    assert np.array_equal(ri, gi) and np.array_equal(rd, gd)


def test_compat_search_from_two_streams(zv, oracle):
    """Batched compat searches launched on two streams of one index (they share the index's tie-flag scratch: each launch
    pair waits for the previous one's marker) return what a single stream returns -- on data where many queries take the
    equal-minima fallback."""
    import torch
    from zvdb_amd.index import Index
    rng = np.random.default_rng(5)
    V = np.floor(rng.random((4000, 4), dtype=np.float32) * 3).astype(np.float32)
    h = zv.HNSW(8, 200)
    h.insert_many(V, np.zeros(len(V), dtype=np.int32))
    idx = Index._from_handle(h._L.zvdb_host_hnsw_handle(h._p))
    dev = torch.device("cuda", 0)
    nq, k, d = 3000, 10, 4
    qa = torch.from_numpy(V[:nq].copy()).to(dev)
    qb = torch.from_numpy((V[500:500 + nq] + np.float32(0.25)).copy()).to(dev)
    def run(q, stream):
        ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
        cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
        idx.search_graph_dev(q.data_ptr(), nq, d, k, 0, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, stream.cuda_stream)
        return ids, dd, cnt
    s0 = torch.cuda.current_stream()
    ra = run(qa, s0); rb = run(qb, s0)
    torch.cuda.synchronize()
    want = [t.cpu().numpy().copy() for t in ra + rb]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(s0); s2.wait_stream(s0)
    for _ in range(10):
        ga = run(qa, s1); gb = run(qb, s2)
        torch.cuda.synchronize()
        got = [t.cpu().numpy() for t in ga + gb]
        assert all(np.array_equal(a, b) for a, b in zip(want, got))


def test_search_many_equals_per_call_search(zv, oracle):
    """HNSW.search_many (one zvdb_hip_search_graph launch for the whole batch; `searchMany` of bindings/hnsw_hip.zig, `search_many` of
    hnsw.hpp) returns, per query, the ids / distances / count of search(): both equal to the oracle's hnsw.zig:194-236, also with
    k larger than the index (fewer than k results, test_hnsw.zig:122-125) and on an empty index."""
    n, d, m, nq = 2_000, 48, 8, 130
    rng = np.random.default_rng(77)
    V = rng.random((n, d), dtype=np.float32)
    Q = rng.random((nq, d), dtype=np.float32)
    ref = oracle.HNSW(m, 200, level_seed=3)
    for r in V:
        ref.insert(r)
    levels = np.array([ref.node_level(i) for i in range(n)], dtype=np.int32)
    h = zv.HNSW(m, 200)
    h.insert_many(V, levels)
    for k in (1, 10, 64):
        ids, dist, cnt = h.search_many(Q, k)
        assert ids.shape == (nq, k) and dist.shape == (nq, k) and cnt.shape == (nq,)
        for qi in range(nq):
            ri, rd = ref.search(Q[qi], k)
            c = int(cnt[qi])
            assert c == len(ri) and np.array_equal(ids[qi, :c], ri) and np.array_equal(dist[qi, :c], rd), (k, qi)
            assert np.all(ids[qi, c:] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(np.isinf(dist[qi, c:]))
            if qi % 13 == 0:
                pi, pd = h.search_ids(Q[qi], k)
                assert np.array_equal(pi, ids[qi, :c]) and np.array_equal(pd, dist[qi, :c])
    small = zv.HNSW(m, 200)
    small.insert_many(V[:5], np.zeros(5, dtype=np.int32))
    ids, dist, cnt = small.search_many(Q[:7], 10)
    r5 = oracle.HNSW(m, 200, level_seed=3)
    for r in V[:5]:
        r5.insert(r, 0)
    for qi in range(7):
        ri, rd = r5.search(Q[qi], 10)
        assert int(cnt[qi]) == len(ri) <= 5 and np.array_equal(ids[qi, :len(ri)], ri)
    empty = zv.HNSW(m, 200)
    ids, dist, cnt = empty.search_many(Q[:3], 4)
    assert np.all(cnt == 0) and np.all(ids == np.uint64(0xFFFFFFFFFFFFFFFF))
