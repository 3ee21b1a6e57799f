"""GPU tests of the bulk graph (zvdb_hip_build_graph, SURVEY 8f rank 1 + 3) against its specification, the oracle
(orc_bulk_level / orc_graph_select / orc_graph_build_layer0 / orc_hnsw_descend / orc_hnsw_search_ef_ex): levels, EVERY list of
EVERY layer, the entry node, each query's descent result, and the beam search from it -- ids, distances, counters bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zv():
    import zvdb_amd
    assert zvdb_amd.device_count() > 0
    return zvdb_amd


@pytest.mark.parametrize("n,d,M,efc,metric,ties", [
    (1500, 24, 8, 40, "l2", False),
    (3000, 64, 16, 64, "l2", False),
    (2000, 100, 4, 96, "l2", False),
    (1200, 8, 6, 30, "l2", True),        # grid data: exactly equal distances everywhere
    (2500, 48, 8, 48, "dot", False),
    (40, 16, 16, 64, "l2", False),       # fewer rows than candidates asked for
    (1, 8, 4, 10, "l2", False),
])
def test_bulk_graph_equals_the_oracle_spec(zv, oracle, n, d, M, efc, metric, ties):
    m = {"l2": zv.METRIC_L2SQ, "dot": zv.METRIC_DOT}[metric]
    om = 0 if metric == "l2" else 1
    rng = np.random.default_rng(n + d)
    V = rng.random((n, d), dtype=np.float32) - np.float32(0.0 if metric == "l2" else 0.5)
    if ties:
        V = np.floor(V * 3).astype(np.float32)
    seed = 0xABC + n
    idx = zv.Index(metric=m, m=2 * M, graph_stride=2 * M, flags=zv.FLAG_BF16_MIRROR)
    idx.add(V)
    idx.build_graph(M, efc, zv.BF_F32, seed)
    ref, levels, top = oracle.build_bulk_graph(V, M, efc, seed, metric=om)
    e, ml = idx.graph_info()
    assert (e, ml) == (top, int(levels.max()))
    for i in range(n):
        assert [int(x) for x in idx.get_neighbors(i)] == [int(x) for x in ref.neighbors(i, 0)], ("layer 0", i)
        for L in range(1, int(levels[i]) + 1):
            assert [int(x) for x in idx.get_neighbors_layer(i, L)] == [int(x) for x in ref.neighbors(i, L)], (L, i)
    if n < 2:
        return
    Q = rng.random((40, d), dtype=np.float32) - np.float32(0.0 if metric == "l2" else 0.5)
    if ties:
        Q = np.floor(Q * 3).astype(np.float32)
    for k, ef in ((10, 10), (10, 64), (25, 128)):
        ids, dist, cnt, (ev, pp) = idx.search_graph(Q, k, ef=ef, entry=zv.ENTRY_DESCEND, with_stats=True)
        ref.reset_counters()
        for qi, q in enumerate(Q):
            ent = ref.descend(q, om, top)
            ri, rd = ref.search_ef(q, k, ef, om, ent)
            assert cnt[qi] == len(ri), (k, ef, qi)
            assert np.array_equal(ids[qi, :len(ri)], ri) and np.array_equal(dist[qi, :len(ri)], rd), (k, ef, qi)
        c = ref.counters()
        assert (ev, pp) == (c.search_dist_evals, c.search_pops), (k, ef)


@pytest.mark.parametrize("width", [2, 3, 4])
@pytest.mark.parametrize("M,bf16rows,metric", [(8, False, "l2"), (16, True, "l2"), (8, False, "dot")])
def test_search_width_equals_the_oracle_spec(zv, oracle, width, M, bf16rows, metric):
    """ZVDB_OPT_EF_SEARCH_WIDTH = w: the beam search takes the first w unexpanded entries per step (spec: orc_hnsw_search_ef_w) --
    ids, distances, counts and the E / P counters bit for bit, entered by descent on the bulk graph: layer-0 stride 16 (w lists side
    by side in one window) and 32 (w = 3, 4 fall back to list by list), fp32 and bf16 rows, both metrics, small and large ef (the
    LDS-beam kernel beyond 1,024), and with the visited table forced tiny (retry + net passes)."""
    n, d, efc = 4000, 40, 48
    m = {"l2": zv.METRIC_L2SQ, "dot": zv.METRIC_DOT}[metric]
    om = 0 if metric == "l2" else 1
    rng = np.random.default_rng(100 * width + M)
    V = rng.random((n, d), dtype=np.float32) - np.float32(0.0 if metric == "l2" else 0.5)
    if bf16rows:
        V = oracle.bf16_round(V)
    idx = zv.Index(metric=m, m=2 * M, graph_stride=2 * M, flags=zv.FLAG_BF16_ROWS if bf16rows else zv.FLAG_BF16_MIRROR)
    idx.add(V)
    idx.build_graph(M, efc, zv.BF_I8, 21)
    ref, levels, top = oracle.build_bulk_graph(V, M, efc, 21, metric=om)
    idx.set_option(zv.OPT_EF_SEARCH_WIDTH, width)
    Q = rng.random((48, d), dtype=np.float32) - np.float32(0.0 if metric == "l2" else 0.5)
    for k, ef, tune in ((10, 16, {}), (10, 128, {}), (50, 300, {}), (10, 1100, {}), (10, 96, {"ZVDB_EF_TWO_PASS": "t8"}),
                        (10, 96, {"ZVDB_EF_HASH_BITS": "8", "ZVDB_EF_TWO_PASS": "0"}), (10, 64, {"ZVDB_EF_BEAM": "lds"})):
        with zv.tuned(**tune):
            ids, dist, cnt, (ev, pp) = idx.search_graph(Q, k, ef=ef, entry=zv.ENTRY_DESCEND, with_stats=True)
        ref.reset_counters()
        for qi, q in enumerate(Q):
            ent = ref.descend(q, om, top)
            ri, rd = ref.search_ef(q, k, ef, om, ent, width)
            assert cnt[qi] == len(ri), (k, ef, tune, qi)
            assert np.array_equal(ids[qi, :len(ri)], ri) and np.array_equal(dist[qi, :len(ri)], rd), (k, ef, tune, qi)
        c = ref.counters()
        assert (ev, pp) == (c.search_dist_evals, c.search_pops), (k, ef, tune)


@pytest.mark.parametrize("width,M,bf16rows", [(1, 8, False), (4, 8, True), (4, 16, False)])
def test_large_beams_packed_visited_table_gather_merge_wide_workgroups(zv, oracle, width, M, bf16rows):
    """ef > 256 (register beam of 8 / 16 blocks): the visited set in 21-bit entries, three to a word, behind a bijection of the ids
    (exact membership), the window's candidates merged by gather from the LDS copy of the beam, 2 / 4 / 8 waves per query -- ids,
    distances and the E / P counters equal the oracle's search in every combination; also with the packed table switched off, with
    MORE regions than the row count needs (ZVDB_EF_PACKED_R: 2^4 regions of a forced 768-unit table are 24 words each -- regions
    fill up and the query is handed to the next pass) and with the table forced tiny (overflow -> retry -> net)."""
    n, d, efc = 4000, 24, 48
    rng = np.random.default_rng(7 * width + M)
    V = rng.random((n, d), dtype=np.float32)
    if bf16rows:
        V = oracle.bf16_round(V)
    idx = zv.Index(m=2 * M, graph_stride=2 * M, flags=zv.FLAG_BF16_ROWS if bf16rows else zv.FLAG_BF16_MIRROR)
    idx.add(V)
    idx.build_graph(M, efc, zv.BF_I8, 21)
    ref, levels, top = oracle.build_bulk_graph(V, M, efc, 21)
    idx.set_option(zv.OPT_EF_SEARCH_WIDTH, width)
    Q = rng.random((40, d), dtype=np.float32)
    want = {}
    for k, ef in ((10, 300), (100, 520), (10, 1000)):
        ref.reset_counters()
        res = []
        for q in Q:
            res.append(ref.search_ef(q, k, ef, 0, ref.descend(q, 0, top), width))
        c = ref.counters()
        want[(k, ef)] = (res, (c.search_dist_evals, c.search_pops))
    tunes = [{}, {"ZVDB_EF_PACKED_VISITED": "0"}, {"ZVDB_EF_PACKED_R": "3"}, {"ZVDB_EF_PACKED_R": "4", "ZVDB_EF_TWO_PASS": "s768"},
             {"ZVDB_EF_PACKED_R": "6", "ZVDB_EF_TWO_PASS": "s512"}, {"ZVDB_EF_TWO_PASS": "s320"}, {"ZVDB_EF_TWO_PASS": "t8"}]
    if width > 1:
        tunes += [{"ZVDB_EF_WPQ": "1"}, {"ZVDB_EF_WPQ": "2"}, {"ZVDB_EF_WPQ": "4"}, {"ZVDB_EF_WPQ": "8"}, {"ZVDB_EF_WPQ": "8", "ZVDB_EF_PACKED_R": "5", "ZVDB_EF_TWO_PASS": "s768"}]
    for tune in tunes:
        for (k, ef), (res, counters) in want.items():
            for _ in range(2):  # (the second call runs on the table sized from the first one's fill)
                with zv.tuned(**tune):
                    ids, dist, cnt, (ev, pp) = idx.search_graph(Q, k, ef=ef, entry=zv.ENTRY_DESCEND, with_stats=True)
                for qi, (ri, rd) in enumerate(res):
                    assert cnt[qi] == len(ri), (k, ef, tune, qi)
                    assert np.array_equal(ids[qi, :len(ri)], ri) and np.array_equal(dist[qi, :len(ri)], rd), (k, ef, tune, qi)
                assert (ev, pp) == counters, (k, ef, tune)


def test_entry_descent_fused_into_the_search_equals_the_two_launch_form(zv, oracle):
    """ENTRY_DESCEND: the first pass of the register-beam kernel runs the descent itself; ZVDB_EF_FUSE_DESCENT=0 launches k_descend
    ahead of the search as before -- ids, distances, counts and both counters (descent + search evaluations, pops) are the same, and
    the oracle's, also when a forced tiny visited table sends queries through the retry passes (which read the entry the first pass left)."""
    n, d, M = 6000, 48, 8
    rng = np.random.default_rng(5)
    V = rng.random((n, d), dtype=np.float32)
    idx = zv.Index(m=2 * M, graph_stride=2 * M, flags=zv.FLAG_BF16_MIRROR)
    idx.add(V)
    idx.build_graph(M, 48, zv.BF_I8, 21)
    ref, levels, top = oracle.build_bulk_graph(V, M, 48, 21)
    Q = rng.random((64, d), dtype=np.float32)
    # (ZVDB_EF_DIRECT_NET=1: queries that fill the tight table go straight to the net pass, as searches do once an index knows its fills)
    for k, ef, tune in ((10, 64, {}), (10, 300, {}), (10, 96, {"ZVDB_EF_TWO_PASS": "t8"}), (10, 600, {"ZVDB_EF_TWO_PASS": "s512"}),
                        (10, 96, {"ZVDB_EF_TWO_PASS": "t8", "ZVDB_EF_DIRECT_NET": "1"}), (10, 600, {"ZVDB_EF_TWO_PASS": "s512", "ZVDB_EF_DIRECT_NET": "1"}),
                        (10, 128, {"ZVDB_EF_DIRECT_NET": "1"}), (10, 128, {"ZVDB_EF_DIRECT_NET": "0"})):
        got = {}
        for fuse in ("1", "0"):
            with zv.tuned(ZVDB_EF_FUSE_DESCENT=fuse, **tune):
                got[fuse] = idx.search_graph(Q, k, ef=ef, entry=zv.ENTRY_DESCEND, with_stats=True)
        a, b = got["1"], got["0"]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3], (k, ef, tune)
        for qi, q in enumerate(Q[:16]):
            ri, rd = ref.search_ef(q, k, ef, 0, ref.descend(q, 0, top))
            assert np.array_equal(a[0][qi, :len(ri)], ri) and np.array_equal(a[1][qi, :len(ri)], rd), (k, ef, tune, qi)


def test_search_width_hand_derived_on_the_device(zv, oracle):
    """The paper example of tests/test_oracle_graph_spec.py::test_search_width_hand_derived through the ABI: six points on a line,
    query 4.6, ef = 2 -- width 1 ends on [4, 2] after 3 pops / 4 evaluations, width 2 reaches [5, 4] after 6 / 6."""
    idx = zv.Index(m=2)
    idx.add(np.arange(6, dtype=np.float32).reshape(6, 1))
    adj = np.zeros((6, 2), dtype=np.uint32); deg = np.zeros(6, dtype=np.uint32)
    for i, lst in {0: [1, 2], 1: [0, 3], 2: [0, 4], 3: [1, 5], 4: [2], 5: [3]}.items():
        adj[i, :len(lst)] = lst; deg[i] = len(lst)
    idx.upload_graph(adj, deg)
    q = np.array([[4.6]], dtype=np.float32)
    d = lambda i: oracle.distance(q[0], np.array([i], dtype=np.float32))
    for width, want, pops, evals in ((1, [4, 2], 3, 4), (2, [5, 4], 6, 6), (4, [5, 4], 6, 6)):
        idx.set_option(zv.OPT_EF_SEARCH_WIDTH, width)
        for tune in ({}, {"ZVDB_EF_BEAM": "lds"}):
            with zv.tuned(**tune):
                ids, dist, cnt, (ev, pp) = idx.search_graph(q, 2, ef=2, with_stats=True)
            assert cnt[0] == 2 and [int(x) for x in ids[0]] == want and list(dist[0]) == [d(want[0]), d(want[1])], (width, tune)
            assert (pp, ev) == (pops, evals), (width, tune)


def test_bulk_graph_recall_beats_the_knn_graph(zv, oracle):
    """What the builder is for: with the SAME parameter M and at the same ef, the heuristic + reverse-link graph entered by
    descent out-recalls the directed exact M-NN lists entered at node 0 (measured 0.79 vs 0.62 at ef = 128 on 20k x 128
    i.i.d. uniform rows, where no graph does well: tests/manual/graph_recall_probe.py holds the full curves)."""
    n, d, M, k = 20_000, 128, 16, 10
    idx = zv.Index(dim=d, capacity=n, m=2 * M, graph_stride=2 * M, flags=zv.FLAG_BF16_MIRROR)
    idx.add_synthetic(n, d, 0x5EED)
    Q = oracle.uniform_rows(0xC0FFEE, 0, 256, d)
    gt, _, _ = idx.search_bruteforce(Q, k, zv.BF_F32)
    def recall(ids):
        return float(np.mean([len(set(ids[i]) & set(gt[i])) / k for i in range(len(Q))]))
    idx.build_knn_graph(M, zv.BF_I8)
    r_knn = recall(idx.search_graph(Q, k, ef=128)[0])
    idx.build_graph(M, 64, zv.BF_I8, 7)
    r_bulk = recall(idx.search_graph(Q, k, ef=128, entry=zv.ENTRY_DESCEND)[0])
    r_bulk_512 = recall(idx.search_graph(Q, k, ef=512, entry=zv.ENTRY_DESCEND)[0])
    assert r_bulk >= 0.75 and r_bulk > r_knn + 0.1 and r_bulk_512 >= 0.97, (r_knn, r_bulk, r_bulk_512)


@pytest.mark.parametrize("kind", ["compat_fp32", "bulk_bf16_rows"])
def test_on_disk_format_round_trip(zv, oracle, tmp_path, kind):
    """zvdb_hip_save / zvdb_hip_load: the loaded index holds the same rows, the same graph on every layer
    (reference-compat or bulk, with its entry) and returns the same results on every route, bit for bit; the file is the flat
    layout the header documents (magic, 256-byte aligned sections, size = header + sections)."""
    n, d = 5000, 96
    V = oracle.uniform_rows(77, 0, n, d)
    Q = oracle.uniform_rows(78, 0, 32, d)
    path = str(tmp_path / "index.zvdb")
    if kind == "compat_fp32":
        h = zv.HNSW(8, 200)
        ref = oracle.HNSW(8, 200, level_seed=3)
        for r in V[:1500]:
            ref.insert(r)
        lv = np.array([ref.node_level(i) for i in range(1500)], dtype=np.int32)
        h.insert_many(V[:1500], lv)
        idx = h.index
        n = 1500
        searches = [dict(k=10, ef=0, entry=0), dict(k=10, ef=32, entry=0)]
    else:
        idx = zv.Index(m=16, graph_stride=16, flags=zv.FLAG_BF16_ROWS)
        idx.add(V)
        idx.build_graph(8, 40, zv.BF_BF16, 11)
        searches = [dict(k=10, ef=64, entry=zv.ENTRY_DESCEND), dict(k=5, ef=16, entry=0)]
    idx.save(path)
    raw = open(path, "rb").read()
    assert raw[:8] == b"ZVDBHIP1" and len(raw) % 256 == 0
    got = zv.Index.load(path)
    assert got.count() == n and got.dim == d and got.graph_info() == idx.graph_info() and got.storage_info()[0] == idx.storage_info()[0]
    for i in range(0, n, 37):
        assert np.array_equal(got.get_vector(i), idx.get_vector(i))
        assert np.array_equal(got.get_neighbors(i), idx.get_neighbors(i))
    for s in searches:
        a = idx.search_graph(Q, s["k"], ef=s["ef"], entry=s["entry"], with_stats=True)
        b = got.search_graph(Q, s["k"], ef=s["ef"], entry=s["entry"], with_stats=True)
        assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3] == b[3], s
    mode = zv.BF_BF16 if kind != "compat_fp32" else zv.BF_I8
    ai, ad, ac = idx.search_bruteforce(Q, 10, mode)
    bi, bd, bc = got.search_bruteforce(Q, 10, mode)
    assert np.array_equal(ai, bi) and np.array_equal(ad, bd)
    with open(path, "r+b") as f:  # a truncated file is refused, not half-loaded
        f.truncate(len(raw) // 2)
    with pytest.raises(zv.ZvdbError):
        zv.Index.load(path)


@pytest.mark.parametrize("env", [
    {"ZVDB_EF_GLOBAL_VISITED": "1"},                      # safe pass with the visited sets in the index-owned global buffer
    {"ZVDB_EF_GLOBAL_VISITED": "1", "ZVDB_EF_TWO_PASS": "0"},
    {"ZVDB_EF_HASH_BITS": "8", "ZVDB_EF_TWO_PASS": "0"},  # safe LDS table forced tiny: most queries are served by the net (work list + table pool)
    {},
])
def test_ef_search_from_two_streams_shares_the_index_scratch(zv, oracle, env, tuning):
    """ef searches launched on two streams of ONE index share its scratch (global visited tables, the net's work list and table
    pool): each search waits for the previous user's marker (ef_search.hip: ef_scratch_begin / _end).  Results, launched
    back to back on two streams, equal the single-stream results and the oracle's -- in every mode that uses that scratch."""
    import torch
    n, d, m, k, ef, nq = 6000, 32, 8, 10, 48, 1500
    V = oracle.uniform_rows(0x5EED, 0, n, d)
    idx = zv.Index(m=m)
    idx.add(V)
    idx.build_knn_graph(m)
    dev = torch.device("cuda", 0)
    Qa = oracle.uniform_rows(0xC0FFEE, 0, nq, d)
    Qb = oracle.uniform_rows(0xC0FFEE, nq, nq, d)
    qa, qb = torch.from_numpy(Qa).to(dev), torch.from_numpy(Qb).to(dev)
    for key, val in env.items():
        tuning(key, val)
    idx.search_graph_reserve(nq, k, ef)

    def run(q, stream):
        ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
        cnt = torch.empty((nq,), dtype=torch.int32, device=dev); st = torch.zeros((3,), dtype=torch.int64, device=dev)
        with torch.cuda.stream(stream):
            st.zero_()
        idx.search_graph_dev(q.data_ptr(), nq, d, k, ef, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), stream.cuda_stream)
        return ids, dd, cnt, st

    s0 = torch.cuda.current_stream()
    ra = run(qa, s0); rb = run(qb, s0)
    torch.cuda.synchronize()
    want = [t.cpu().numpy().copy() for t in ra + rb]
    assert want[3][2] == 0 and want[7][2] == 0, "a visited set overflowed"
    # the single-stream results are the host entry point's (which the fuzz sweeps hold to the oracle); spot-check the oracle here too
    ref = oracle.HNSW(m, 200)
    for r in V:
        ref.insert(r, 0)
    for i in range(n):
        nb = idx.get_neighbors(i)
        ref.set_neighbors(i, 0, [int(x) for x in nb])
    for qi in range(0, nq, 97):
        ri, rd = ref.search_ef(Qa[qi], k, ef)
        assert np.array_equal(want[0][qi, :len(ri)].astype(np.uint64), ri) and np.array_equal(want[1][qi, :len(ri)], rd)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(s0); s2.wait_stream(s0)
    for _ in range(8):
        ga = run(qa, s1); gb = run(qb, s2)
        torch.cuda.synchronize()
        got = [t.cpu().numpy() for t in ga + gb]
        assert all(np.array_equal(a, b) for a, b in zip(want, got))


def _file_header(raw):
    """The 256-byte header of the on-disk format (api.hip: ZvdbFileHeader) as a dict of (offset, struct format, value)."""
    import struct
    names32 = ["version", "dim", "ld", "ldh", "metric", "m", "ef_construction", "gstride", "flags", "max_level", "bulk_graph", "bulk_top"]
    names64 = ["count", "up_count", "off_rows", "off_norm", "off_adj", "off_deg", "off_lvl", "off_slot", "off_adj_up", "off_deg_up", "file_bytes"]
    out, off = {}, 8
    for nm in names32:
        out[nm] = (off, "<I", struct.unpack_from("<I", raw, off)[0]); off += 4
    for nm in names64:
        out[nm] = (off, "<Q", struct.unpack_from("<Q", raw, off)[0]); off += 8
    return out


def test_load_rejects_corrupt_and_hostile_files(zv, oracle, tmp_path):
    """zvdb_hip_load trusts nothing in the file: header fields outside their domains, a section table that is not the one the
    fields imply, a size that differs from the header's, and -- after the upload -- any index the graph kernels would follow
    out of bounds (degree > stride, neighbour id >= count, level > max_level, upper-layer slots past up_count) all end in
    ZVDB_ERR_INVALID, not in a device fault.  Each case is one field or one word of a valid file changed."""
    import struct
    n, d = 3000, 32
    V = oracle.uniform_rows(91, 0, n, d)
    idx = zv.Index(m=8, graph_stride=16)
    idx.add(V)
    idx.build_graph(8, 32, zv.BF_I8, 5)
    good = str(tmp_path / "good.zvdb")
    idx.save(good)
    raw = open(good, "rb").read()
    hd = _file_header(raw)
    assert hd["count"][2] == n and hd["up_count"][2] > 0 and hd["bulk_graph"][2] == 1
    zv.Index.load(good)  # the unmodified file loads

    def refused(mutated, what):
        p = str(tmp_path / "bad.zvdb")
        with open(p, "wb") as f:
            f.write(mutated)
        with pytest.raises(zv.ZvdbError) as e:
            zv.Index.load(p)
        assert "corrupt index file" in str(e.value) or "not a zvdb-hip index file" in str(e.value), (what, str(e.value))

    def with_field(name, value):
        off, fmt, _ = hd[name]
        b = bytearray(raw)
        struct.pack_into(fmt, b, off, value)
        return bytes(b)

    def with_word(section, index, value):  # one u32 of a section
        b = bytearray(raw)
        struct.pack_into("<I", b, hd[section][2] + 4 * index, value)
        return bytes(b)

    # (1) header domains
    refused(with_field("flags", 0x80), "unknown flag bit")
    refused(with_field("metric", 7), "metric")
    refused(with_field("max_level", 40), "max_level > 31")
    refused(with_field("bulk_top", n), "entry node >= count")
    refused(with_field("gstride", 0), "zero stride")
    refused(with_field("count", 2**40), "absurd row count (overflowing sizes)")
    refused(with_field("up_count", 2**63), "absurd upper-row count")
    # (2) section table / size
    refused(with_field("off_adj", hd["off_adj"][2] + 256), "shifted section")
    refused(with_field("off_rows", 0), "section over the header")
    refused(with_field("file_bytes", hd["file_bytes"][2] + 256), "size mismatch")
    refused(raw + b"\0" * 256, "padded file")
    refused(raw[:len(raw) - 256], "truncated file")
    refused(with_field("count", n - 1), "count that no longer matches the table")
    # (3) graph contents (the header stays consistent: only the kernel can see these)
    refused(with_word("off_deg", 17, 17), "degree > stride")
    refused(with_word("off_adj", 5 * 16 + 0, n + 3), "neighbour id out of range")
    refused(with_word("off_adj", 9 * 16 + 0, 0xFFFFFFFF), "neighbour id out of range (max)")
    refused(with_word("off_lvl", 3, 31), "level above max_level" if hd["max_level"][2] < 31 else "slots out of range")
    refused(with_word("off_deg_up", 0, 99), "upper degree > stride")
    refused(with_word("off_adj_up", 0, n), "upper neighbour id out of range")
    # a node WITH upper layers whose slot points past the upper rows
    lv = np.frombuffer(raw, dtype="<u4", count=n, offset=hd["off_lvl"][2])
    top = int(np.argmax(lv))
    assert lv[top] >= 1
    refused(with_word("off_slot", top, int(hd["up_count"][2])), "upper-layer slot past up_count")


def test_descend_is_refused_once_the_bulk_graph_is_stale(zv, oracle, tmp_path):
    """The bulk graph's upper layers describe the index only as the build left it: after rows are added, or layer 0 is replaced
    (upload_graph / set_neighbors / build_knn_graph), ZVDB_ENTRY_DESCEND is an error instead of a descent through stale layers,
    and graph_info stops reporting the old entry; a rebuild restores it."""
    n, d = 2000, 24
    V = oracle.uniform_rows(5, 0, n + 100, d)
    Q = oracle.uniform_rows(6, 0, 8, d)
    idx = zv.Index(m=8, graph_stride=16)
    idx.add(V[:n])
    idx.build_graph(8, 32, zv.BF_I8, 3)
    entry, maxlev = idx.graph_info()
    idx.search_graph(Q, 5, ef=32, entry=zv.ENTRY_DESCEND)
    idx.add(V[n:])                                            # rows the descent could never reach
    with pytest.raises(zv.ZvdbError):
        idx.search_graph(Q, 5, ef=32, entry=zv.ENTRY_DESCEND)
    assert idx.graph_info()[0] == 0
    idx.search_graph(Q, 5, ef=32, entry=0)                    # a plain entry still works
    path = str(tmp_path / "stale.zvdb")                       # a save / load round trip does not launder the stale graph
    idx.save(path)
    got = zv.Index.load(path)
    assert got.count() == n + 100 and got.graph_info()[0] == 0
    with pytest.raises(zv.ZvdbError):
        got.search_graph(Q, 5, ef=32, entry=zv.ENTRY_DESCEND)
    a = idx.search_graph(Q, 5, ef=32, entry=0)
    b = got.search_graph(Q, 5, ef=32, entry=0)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    with pytest.raises(zv.ZvdbError):                          # a REJECTED list replacement changes nothing -- not the graph's validity either
        idx.build_graph(8, 32, zv.BF_I8, 3) or idx.set_neighbors(7, [n + 5000])
    idx.search_graph(Q, 5, ef=32, entry=zv.ENTRY_DESCEND)
    idx.build_knn_graph(8)                                    # layer 0 replaced under the upper layers
    with pytest.raises(zv.ZvdbError):
        idx.search_graph(Q, 5, ef=32, entry=zv.ENTRY_DESCEND)
    idx.build_graph(8, 32, zv.BF_I8, 3)
    idx.set_neighbors(7, [1, 2, 3])
    with pytest.raises(zv.ZvdbError):
        idx.search_graph(Q, 5, ef=32, entry=zv.ENTRY_DESCEND)


@pytest.mark.parametrize("bf16rows", [False, True])
def test_load_does_not_trust_the_rows_padding_or_the_stored_norms(zv, oracle, tmp_path, bf16rows):
    """Rows sit in the file padded to ld elements and the exact kernels walk whole chunks up to ld: a file whose padding holds
    NaN / garbage, or whose norm section lies, loads into an index that answers exactly like the one that was saved."""
    import struct
    n, d = 4000, 100  # ld = 128 (fp32) / ldh = 128 (bf16): 28 padding elements per row
    V = oracle.uniform_rows(55, 0, n, d)
    Q = oracle.uniform_rows(56, 0, 16, d)
    idx = zv.Index(m=8, graph_stride=8, flags=zv.FLAG_BF16_ROWS if bf16rows else zv.FLAG_BF16_MIRROR)
    idx.add(V)
    idx.build_knn_graph(8)
    path = str(tmp_path / "pad.zvdb")
    idx.save(path)
    raw = bytearray(open(path, "rb").read())
    hd = _file_header(bytes(raw))
    esz, ld = (2, hd["ldh"][2]) if bf16rows else (4, hd["ld"][2])
    assert ld == 128
    rows_off = hd["off_rows"][2]
    for r in range(n):
        for c in range(d, ld):
            if bf16rows:
                struct.pack_into("<H", raw, rows_off + (r * ld + c) * esz, 0x7FC0 if (r + c) % 3 else 0x4120)  # NaN / 10.0
            else:
                struct.pack_into("<f", raw, rows_off + (r * ld + c) * esz, float("nan") if (r + c) % 3 else 10.0)
    for r in range(0, n, 7):
        struct.pack_into("<f", raw, hd["off_norm"][2] + 4 * r, 1.0e9)  # lying norms
    with open(path, "wb") as f:
        f.write(bytes(raw))
    got = zv.Index.load(path)
    for m in (zv.BF_F32, zv.BF_BF16, zv.BF_I8):
        a = idx.search_bruteforce(Q, 10, m)
        b = got.search_bruteforce(Q, 10, m)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), m
    for ef in (0, 32):
        a = idx.search_graph(Q, 10, ef=ef)
        b = got.search_graph(Q, 10, ef=ef)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), ef
    assert np.array_equal(got.get_vector(17), idx.get_vector(17))


def test_load_accepts_every_index_create_accepts(zv, oracle, tmp_path):
    """m may exceed the device stride (zvdb_hip_create allows it; the bulk build's upper-layer sub-indexes are such): the loader's
    header checks follow create's domains, not a narrower guess."""
    V = oracle.uniform_rows(3, 0, 800, 16)
    idx = zv.Index(m=16, graph_stride=8)
    idx.add(V)
    idx.build_knn_graph(8)
    path = str(tmp_path / "m_gt_stride.zvdb")
    idx.save(path)
    got = zv.Index.load(path)
    Q = oracle.uniform_rows(4, 0, 8, 16)
    a = idx.search_graph(Q, 5, ef=16)
    b = got.search_graph(Q, 5, ef=16)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
