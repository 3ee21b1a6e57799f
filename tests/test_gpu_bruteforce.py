"""GPU parity tests of the brute-force path (a10; no reference symbol -- pinned to the oracle's
exact (distance, id) top-k): MFMA screen (fp32 / bf16) + fused threshold top-k' + exact re-rank.
Bit-exact ids AND distances (the re-rank recomputes the reference arithmetic)."""
import numpy as np
import pytest
from conftest import tuning_prelude

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zv():
    import zvdb_amd
    assert zvdb_amd.device_count() > 0
    return zvdb_amd


def MODES(zv):
    return {"f32": zv.BF_F32, "bf16": zv.BF_BF16, "i8": zv.BF_I8, "fp4": zv.BF_FP4}


def _index(zv, V, **kw):
    idx = zv.Index(flags=zv.FLAG_BF16_MIRROR, **kw)
    idx.add(V)
    return idx


@pytest.mark.parametrize("n,d,nq,k", [(400, 64, 8, 10), (5000, 128, 300, 10), (20000, 96, 130, 32),
                                      (3000, 3, 64, 5), (9000, 40, 17, 48), (100, 128, 5, 10),
                                      (1, 16, 3, 4), (129, 256, 1, 1)])
@pytest.mark.parametrize("mode", ["f32", "bf16", "i8"])
def test_bruteforce_bit_exact_vs_oracle(zv, oracle, n, d, nq, k, mode):
    V = oracle.uniform_rows(1000 + n, 0, n, d)
    Q = oracle.uniform_rows(2000 + n, 0, nq, d)
    idx = _index(zv, V)
    ids, dist, cnt = idx.search_bruteforce(Q, k, MODES(zv)[mode])
    ri, rd, rc = oracle.bruteforce_topk(V, Q, k)
    assert np.array_equal(cnt, rc)
    for qi in range(nq):
        c = int(rc[qi])
        assert np.array_equal(ids[qi, :c], ri[qi, :c]), (qi, ids[qi], ri[qi])
        assert np.array_equal(dist[qi, :c], rd[qi, :c])
        assert np.all(ids[qi, c:] == np.uint64(0xFFFFFFFFFFFFFFFF))


@pytest.mark.parametrize("mode", ["f32", "bf16", "i8", "fp4"])
def test_bruteforce_ties_by_lower_id(zv, oracle, mode):
    """Duplicated rows and grid data: exactly equal distances must come back lowest id first."""
    rng = np.random.default_rng(4)
    nb = 24 if mode == "fp4" else 300  # fp4: 60 rows <= k' = 64, so every row reaches the exact re-rank
    base = rng.integers(0, 4, size=(nb, 8)).astype(np.float32)
    V = np.concatenate([base, base, base[:nb // 3]], axis=0)  # every row appears 2-3 times
    Q = rng.integers(0, 4, size=(40, 8)).astype(np.float32)
    idx = _index(zv, V)
    ids, dist, cnt = idx.search_bruteforce(Q, 20, MODES(zv)[mode])
    ri, rd, _ = oracle.bruteforce_topk(V, Q, 20)
    assert np.array_equal(ids, ri) and np.array_equal(dist, rd)


def test_bruteforce_adversarial_order_forces_prunes(zv, oracle):
    """Rows sorted by DEcreasing distance to the query: every row passes the running threshold,
    so the in-kernel prune runs every tile (a rare data-dependent path, tested on purpose)."""
    V = oracle.uniform_rows(77, 0, 6000, 32)
    q = oracle.uniform_rows(78, 0, 1, 32)[0]
    d = np.array([oracle.distance(q, v) for v in V])
    V = V[np.argsort(-d, kind="stable")]
    Q = np.stack([q, V[0], V[-1]])
    idx = _index(zv, V)
    for mode in (zv.BF_F32, zv.BF_BF16, zv.BF_I8, zv.BF_FP4):
        ids, dist, _ = idx.search_bruteforce(Q, 10, mode)
        ri, rd, _ = oracle.bruteforce_topk(V, Q, 10)
        assert np.array_equal(ids, ri) and np.array_equal(dist, rd)


def test_bruteforce_large_self_consistency(zv, oracle):
    """BASELINE config 2 shape scaled to test size (200k x 128, batch 256): the bf16 screen and the
    fp32 screen must agree on every id (recall 1.0 against each other), distances exact, sorted."""
    idx = zv.Index(flags=zv.FLAG_BF16_MIRROR)
    idx.add_synthetic(200_000, 128, 0x5EED)
    Q = oracle.uniform_rows(0xC0FFEE, 0, 256, 128)
    i32, d32, _ = idx.search_bruteforce(Q, 10, zv.BF_F32)
    i16, d16, _ = idx.search_bruteforce(Q, 10, zv.BF_BF16)
    assert np.array_equal(i32, i16) and np.array_equal(d32, d16)
    i8, d8, _ = idx.search_bruteforce(Q, 10, zv.BF_I8)
    assert np.array_equal(i32, i8) and np.array_equal(d32, d8)
    assert np.all(np.diff(d32, axis=1) >= 0)
    # spot-check 8 queries against the oracle on the same synthetic stream
    V = oracle.uniform_rows(0x5EED, 0, 200_000, 128)
    assert np.array_equal(idx.get_vector(123_456), V[123_456])
    ri, rd, _ = oracle.bruteforce_topk(V, Q[:8], 10)
    assert np.array_equal(i32[:8], ri) and np.array_equal(d32[:8], rd)


@pytest.mark.parametrize("n,d,nq,k", [(400, 64, 8, 10), (100, 128, 5, 10), (1, 16, 3, 4), (129, 256, 1, 1), (60, 300, 70, 48)])
def test_bruteforce_fp4_exact_when_corpus_fits_the_survivor_list(zv, oracle, n, d, nq, k):
    """fp4 screen, N <= a few x k' = 64 survivors: (almost) every row reaches the exact re-rank, so the result
    must equal the oracle's bit for bit -- checks the operand packing, the MFMA lane map (a wrong map scrambles
    the keys and loses neighbours even here for n = 400), norms, ragged tiles and padding."""
    V = oracle.uniform_rows(3000 + n, 0, n, d)
    Q = oracle.uniform_rows(4000 + n, 0, nq, d)
    idx = _index(zv, V)
    ids, dist, cnt = idx.search_bruteforce(Q, k, zv.BF_FP4)
    ri, rd, rc = oracle.bruteforce_topk(V, Q, k)
    assert np.array_equal(cnt, rc)
    for qi in range(nq):
        c = int(rc[qi])
        assert np.array_equal(ids[qi, :c], ri[qi, :c]), (qi, ids[qi], ri[qi])
        assert np.array_equal(dist[qi, :c], rd[qi, :c])


@pytest.mark.parametrize("n,d,nq", [(20000, 96, 130), (50000, 768, 64), (200_000, 256, 32)])
def test_bruteforce_fp4_recall_and_exact_distances(zv, oracle, n, d, nq):
    """The 4-bit screen is approximate: recall@10 against the exact route is MEASURED (bar: >= 0.95, the
    headline metric's bar); whatever it returns carries the reference's exact distance, ascending, ties by id,
    and every returned id that is a true neighbour sits at the oracle's distance."""
    idx = zv.Index()
    idx.add_synthetic(n, d, 0x5EED)
    Q = oracle.uniform_rows(0xC0FFEE, 0, nq, d)
    ei, ed, _ = idx.search_bruteforce(Q, 10, zv.BF_F32)
    ai, ad, ac = idx.search_bruteforce(Q, 10, zv.BF_FP4)
    assert np.all(ac == 10)
    rec = np.mean([len(set(ai[i]) & set(ei[i])) / 10 for i in range(nq)])
    assert rec >= 0.95, rec
    assert np.all(np.diff(ad, axis=1) >= 0)
    for qi in range(0, nq, 7):
        for j in range(10):
            assert oracle.distance(Q[qi], idx.get_vector(int(ai[qi, j]))) == ad[qi, j]


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_bruteforce_fp4_guessed_thresholds_verified_and_repaired(zv, oracle, tmp_path, metric):
    """Large corpus, fp4 stream kernel: ONE scan against thresholds GUESSED from sample moments, verified by the
    merge, repaired by a gated exact re-scan.  All of these must return the same ids and distances, because each
    ends with the exact k' best (fp4 key, id) pairs per query before the exact re-rank:
      default (guess holds) == cascade with learned thresholds (ZVDB_BF_GUESS=0) == staged kernel (ZVDB_BF_FP4_KERNEL=v6)
      == the 16x16x128 MFMA shape of the stream kernel (ZVDB_BF_FP4_SHAPE=16, different mirror layout)
      == guess forced far too tight (ZSHIFT=-3: every query is repaired) == guess far too loose (ZSHIFT=+2: prune rounds)."""
    import subprocess, sys, os
    code = (
        "import sys, numpy as np, oracle, zvdb_amd as zv\n"
        "idx = zv.Index(metric=zv.METRIC_DOT if sys.argv[2] == 'dot' else zv.METRIC_L2SQ); idx.add_synthetic(600_000, 96, 0x5EED)\n"
        "Q = oracle.uniform_rows(0xC0FFEE, 0, 300, 96)\n"
        "i, d, c = idx.search_bruteforce(Q, 10, zv.BF_FP4)\n"
        "e, ed, _ = idx.search_bruteforce(Q, 10, zv.BF_F32)\n"
        "i8, d8, _ = idx.search_bruteforce(Q, 10, zv.BF_I8)\n"
        # rows wider than the bf16 stream kernel takes: the STAGED kernel's bf16 and fp32 screens (guessed thresholds from a row sample)
        "del idx; wide = zv.Index(metric=zv.METRIC_DOT if sys.argv[2] == 'dot' else zv.METRIC_L2SQ, flags=zv.FLAG_BF16_MIRROR); wide.add_synthetic(540_000, 800, 0x5EED)\n"
        "Qw = oracle.uniform_rows(0xC0FFEE, 0, 40, 800)\n"
        "wb, wbd, _ = wide.search_bruteforce(Qw, 10, zv.BF_BF16)\n"
        "wf, wfd, _ = wide.search_bruteforce(Qw, 10, zv.BF_F32)\n"
        "np.savez(sys.argv[1], i=i, d=d, c=c, e=e, ed=ed, i8=i8, d8=d8, wb=wb, wbd=wbd, wf=wf, wfd=wfd)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for name, env in {"guess": {}, "cascade": {"ZVDB_BF_GUESS": "0"}, "staged": {"ZVDB_BF_FP4_KERNEL": "v6"}, "shape16": {"ZVDB_BF_FP4_SHAPE": "16"},
                      "i8staged": {"ZVDB_BF_I8_KERNEL": "v6"},
                      "tight": {"ZVDB_BF_GUESS_ZSHIFT": "-3"}, "loose": {"ZVDB_BF_GUESS_ZSHIFT": "2"},
                      # r5: under the certificate a deficient guess is repaired by the exact scan ("tight" above); the gated re-scan +
                      # merge pair it replaced stays behind a switch (and serves uncertified / fp4 calls), as does the separate prep launch
                      "tight_gated": {"ZVDB_BF_GUESS_ZSHIFT": "-3", "ZVDB_BF_CERT_DEFER": "0"}, "prep_launch": {"ZVDB_BF_FOLD_PREP": "0"}}.items():
        out = tmp_path / (name + ".npz")
        r = subprocess.run([sys.executable, "-c", tuning_prelude(env) + code, str(out), metric], cwd=root, capture_output=True, text=True)
        assert r.returncode == 0, name + ": " + r.stdout + r.stderr
        res[name] = np.load(out)
    base = res["cascade"]
    rec = np.mean([len(set(base["i"][q]) & set(base["e"][q])) / 10 for q in range(300)])
    assert rec >= (0.95 if metric == "l2" else 0.5), rec  # (dot on all-positive U[0,1) data ranks by norm: a hard case for 4 bits)
    for name in ("guess", "staged", "shape16", "tight", "loose", "tight_gated", "prep_launch"):
        assert np.array_equal(res[name]["i"], base["i"]), name
        assert np.array_equal(res[name]["d"], base["d"]), name
        assert np.array_equal(res[name]["c"], base["c"]), name
    # the int8 route rides the same pipeline (stream kernel + guessed thresholds; "cascade" and "i8staged" = the staged
    # kernel with learned thresholds): every variant must return the EXACT top-k
    for name in res:  # (both metrics: stream kernel == staged kernel, bit for bit)
        assert np.array_equal(res[name]["i8"], base["i8"]), name
        assert np.array_equal(res[name]["d8"], base["d8"]), name
    if metric == "l2":
        assert np.array_equal(base["i8"], base["e"]) and np.array_equal(base["d8"], base["ed"])
    # the staged kernel's float screens ride the pipeline too (thresholds guessed from a 4,096-row sample, k_sample_guess): the
    # fp32 screen on the 96-d shard and both float screens on 800-d rows return the cascade's results under every variant
    for name in res:
        for key in ("e", "ed", "wb", "wbd", "wf", "wfd"):
            assert np.array_equal(res[name][key], base[key]), (name, key)
    assert np.array_equal(base["wb"], base["wf"]) and np.array_equal(base["wbd"], base["wfd"])


def test_bruteforce_fp4_ragged_large_and_incremental(zv, oracle):
    """Stream kernel + guessed thresholds on a corpus that is not a multiple of anything (rows, queries), searched,
    GROWN (the lazy fp4 mirror and its value range are rebuilt) and searched again; k = 48 (the mode's maximum).
    Checks: recall against the exact route, exact ascending distances, no id past the row count, counts."""
    n0, d, nq = 530_003, 200, 261
    idx = zv.Index()
    idx.add_synthetic(n0, d, 0x5EED)
    Q = oracle.uniform_rows(0xC0FFEE, 0, nq, d)
    for n, k in ((n0, 10), (n0 + 70_001, 48)):
        if n > n0:
            idx.add_synthetic(n - n0, d, 0x5EED, first_row=n0)
        assert idx.count() == n
        ei, ed, _ = idx.search_bruteforce(Q, k, zv.BF_F32)
        ai, ad, ac = idx.search_bruteforce(Q, k, zv.BF_FP4)
        assert np.all(ac == k) and np.all(ai < n)
        rec = np.mean([len(set(ai[i]) & set(ei[i])) / k for i in range(nq)])
        assert rec >= (0.95 if k == 10 else 0.93), (n, k, rec)  # (k = 48 runs with k' = 128 survivors: 0.97 measured; 0.87 with 64)
        assert np.all(np.diff(ad, axis=1) >= 0)
        for qi in (0, nq - 1):
            for j in range(0, k, 7):
                assert oracle.distance(Q[qi], idx.get_vector(int(ai[qi, j]))) == ad[qi, j]
        # the int8 stream kernel (KS = 12 at d = 200) on the same ragged / grown corpus: the exact top-k
        i8, d8, c8 = idx.search_bruteforce(Q, k, zv.BF_I8)
        assert np.array_equal(i8, ei) and np.array_equal(d8, ed) and np.all(c8 == k)


@pytest.mark.parametrize("k", [64, 100, 112, 113, 200, 240])
def test_bruteforce_k_up_to_240_on_the_stream_pipelines(zv, oracle, k):
    """k > 48 (the reference's own benchmark goes to k = 100: single_threaded_benchmarks.zig:28-33): the stream pipelines
    switch to k' = 128 survivors (two entries per lane in the selects, 128-wide lists / merge / re-rank) and, for k > 112
    (fp4: k > 64), to k' = 256 (four per lane).  int8 route: the exact top-k, bit for bit against the oracle; fp4 route:
    measured recall, exact ascending distances."""
    n, d, nq = 600_000, 96, 300
    idx = zv.Index()
    idx.add_synthetic(n, d, 0x5EED)
    Q = oracle.uniform_rows(0xC0FFEE, 0, nq, d)
    i8, d8, c8 = idx.search_bruteforce(Q, k, zv.BF_I8)
    assert np.all(c8 == k) and np.all(np.diff(d8, axis=1) >= 0)
    V = oracle.uniform_rows(0x5EED, 0, n, d)
    sel = [0, 1, 150, 298, 299]
    ri, rd, _ = oracle.bruteforce_topk(V, Q[sel], k)
    assert np.array_equal(i8[sel], ri) and np.array_equal(d8[sel], rd)
    a4, d4, c4 = idx.search_bruteforce(Q, k, zv.BF_FP4)
    assert np.all(c4 == k) and np.all(np.diff(d4, axis=1) >= 0)
    rec = np.mean([len(set(a4[i]) & set(i8[i])) / k for i in range(nq)])
    # (k = 64 of 128 and k = 100 of 256 survivors leave slack; k = 112 .. 240 of 256 leave little for a 4-bit screen)
    assert rec >= (0.93 if k <= 100 else 0.75), (k, rec)
    for qi in sel:
        for j in range(0, k, 9):
            assert oracle.distance(Q[qi], V[int(a4[qi, j])]) == d4[qi, j]
    with pytest.raises(zv.ZvdbError):
        idx.search_bruteforce(Q, 241, zv.BF_I8)


@pytest.mark.parametrize("n,d,nq,k", [(40_000, 96, 70, 100), (300_000, 1024, 40, 100), (9000, 40, 17, 57), (70, 32, 5, 112), (2000, 2048, 9, 64)])
def test_bruteforce_k_up_to_112_on_the_staged_routes(zv, oracle, n, d, nq, k):
    """k > 48 (56) on the staged kernel (fp32 / bf16 screens, int8 beyond 768 dimensions, fp4 beyond 1536): the same
    k' = 128 lists (KPV = 128 instantiation of the staged scan + the cascade's merges).  The reference's benchmark grid
    holds dims up to 1024 x k up to 100.  Exact screens: bit for bit against the oracle on sampled queries."""
    V = oracle.uniform_rows(9000 + n, 0, n, d)
    Q = oracle.uniform_rows(9500 + n, 0, nq, d)
    idx = _index(zv, V)
    sel = sorted(set([0, nq // 2, nq - 1]))
    ri, rd, rc = oracle.bruteforce_topk(V, Q[sel], k)
    for mode in (zv.BF_F32, zv.BF_BF16, zv.BF_I8):
        ids, dist, cnt = idx.search_bruteforce(Q, k, mode)
        assert np.all(cnt == min(k, n))
        for j, qi in enumerate(sel):
            c = int(rc[j])
            assert np.array_equal(ids[qi, :c], ri[j, :c]), (mode, qi)
            assert np.array_equal(dist[qi, :c], rd[j, :c]), (mode, qi)
    a4, d4, c4 = idx.search_bruteforce(Q, k, zv.BF_FP4)
    assert np.all(c4 == min(k, n))
    for j, qi in enumerate(sel):
        c = int(rc[j])
        assert np.all(np.diff(d4[qi, :c]) >= 0)
        for e in range(0, c, 13):
            assert oracle.distance(Q[qi], V[int(a4[qi, e])]) == d4[qi, e]
    if d > 128:
        with pytest.raises(zv.ZvdbError):
            idx.search_bruteforce(Q, 113, zv.BF_F32)
    else:  # (fp32 rows of <= 128 dimensions have a stream route of their own since r3: k <= 240, as one flooding scan on small shards)
        kk = min(200, n)
        ri2, rd2, rc2 = oracle.bruteforce_topk(V, Q[sel], kk)
        ids, dist, cnt = idx.search_bruteforce(Q, kk, zv.BF_F32)
        assert np.all(cnt == kk)
        for j, qi in enumerate(sel):
            assert np.array_equal(ids[qi], ri2[j]) and np.array_equal(dist[qi], rd2[j]), qi
        with pytest.raises(zv.ZvdbError):
            idx.search_bruteforce(Q, 241, zv.BF_F32)


@pytest.mark.parametrize("n,d,nq,k", [(5000, 64, 40, 100), (129, 200, 3, 112), (70, 32, 5, 100), (30000, 96, 300, 64),
                                      (5000, 64, 40, 200), (300, 32, 5, 240), (20000, 768, 20, 150)])
def test_bruteforce_k_up_to_240_on_small_shards(zv, oracle, n, d, nq, k):
    """k > 48 on shards too small for the guessed-threshold pipeline: one flooding scan of the stream kernel with 128-wide lists
    (no bound to start with; the prune rounds learn it).  int8 route: bit-exact vs the oracle, also when k > N."""
    V = oracle.uniform_rows(7000 + n, 0, n, d)
    Q = oracle.uniform_rows(8000 + n, 0, nq, d)
    idx = _index(zv, V)
    ids, dist, cnt = idx.search_bruteforce(Q, k, zv.BF_I8)
    ri, rd, rc = oracle.bruteforce_topk(V, Q, k)
    assert np.array_equal(cnt, rc)
    for qi in range(nq):
        c = int(rc[qi])
        assert np.array_equal(ids[qi, :c], ri[qi, :c]), (qi, ids[qi][:8], ri[qi][:8])
        assert np.array_equal(dist[qi, :c], rd[qi, :c])
        assert np.all(ids[qi, c:] == np.uint64(0xFFFFFFFFFFFFFFFF))
    a4, d4, c4 = idx.search_bruteforce(Q, k, zv.BF_FP4)  # fp4: counts, order, exact distances of whatever it returns
    assert np.array_equal(c4, rc)
    for qi in range(nq):
        c = int(rc[qi])
        assert np.all(np.diff(d4[qi, :c]) >= 0)
        for j in range(0, c, 11):
            assert oracle.distance(Q[qi], V[int(a4[qi, j])]) == d4[qi, j]


@pytest.mark.parametrize("d", [1024, 1536])
def test_bruteforce_fp4_wide_rows_stream_equals_staged(zv, oracle, tmp_path, d):
    """768 < dim <= 1536: the fp4 screen runs on the 128-query stream kernel (bf_i8_stream.hip, MODE 3; KS = 18 / 24) with
    guessed + verified thresholds.  It must return exactly what the staged fp4 kernel returns (both end with the exact k'
    best (fp4 key, id) pairs), and hold the recall bar against the exact route."""
    import subprocess, sys, os
    code = (
        "import sys, numpy as np, oracle, zvdb_amd as zv\n"
        "d = int(sys.argv[2]); idx = zv.Index(); idx.add_synthetic(560_000, d, 0x5EED)\n"
        "Q = oracle.uniform_rows(0xC0FFEE, 0, 70, d)\n"
        "i, dd, c = idx.search_bruteforce(Q, 10, zv.BF_FP4)\n"
        "e, ed, _ = idx.search_bruteforce(Q, 10, zv.BF_F32)\n"
        "np.savez(sys.argv[1], i=i, d=dd, c=c, e=e)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for name, env in {"stream": {}, "staged": {"ZVDB_BF_FP4_KERNEL": "v6"}, "repair": {"ZVDB_BF_GUESS_ZSHIFT": "-3"}}.items():
        out = tmp_path / (name + ".npz")
        r = subprocess.run([sys.executable, "-c", tuning_prelude(env) + code, str(out), str(d)], cwd=root, capture_output=True, text=True)
        assert r.returncode == 0, name + ": " + r.stdout + r.stderr
        res[name] = np.load(out)
    base = res["staged"]
    rec = np.mean([len(set(base["i"][q]) & set(base["e"][q])) / 10 for q in range(70)])
    assert rec >= 0.95, rec
    for name in ("stream", "repair"):
        assert np.array_equal(res[name]["i"], base["i"]) and np.array_equal(res[name]["d"], base["d"]), name


@pytest.mark.parametrize("metric", ["dot", "cosine"])
def test_bruteforce_fp4_dot_and_cosine(zv, oracle, metric):
    """fp4 screen under the inner-product metrics (symmetric range, no norm term): whatever it returns is exactly what
    the exact route computes for those ids; recall against the exact route.  The ranking does not depend on the query's
    length but the quantiser does: queries are rescaled to the rows' spread before they are quantised, so queries of any
    scale (cosine rows are normalised at insert, queries arrive as they are) screen equally well."""
    m = zv.METRIC_DOT if metric == "dot" else zv.METRIC_COSINE
    V = oracle.uniform_rows(5, 0, 40_000, 128) - 0.5
    Q = oracle.uniform_rows(6, 0, 64, 128) - 0.5
    idx = zv.Index(metric=m)
    idx.add(V)
    ei, ed, _ = idx.search_bruteforce(Q, 10, zv.BF_F32)
    ai, ad, ac = idx.search_bruteforce(Q, 10, zv.BF_FP4)
    assert np.all(ac == 10) and np.all(np.diff(ad, axis=1) >= 0)
    rec = np.mean([len(set(ai[i]) & set(ei[i])) / 10 for i in range(64)])
    assert rec >= 0.97, rec
    for scale in (1e-3, 37.0):  # the same neighbours whatever the queries' length (fp4 and int8 screens)
        si, _, _ = idx.search_bruteforce(Q * np.float32(scale), 10, zv.BF_FP4)
        assert np.mean([len(set(si[i]) & set(ei[i])) / 10 for i in range(64)]) >= 0.97, scale
        s8, _, _ = idx.search_bruteforce(Q * np.float32(scale), 10, zv.BF_I8)
        assert np.mean([len(set(s8[i]) & set(ei[i])) / 10 for i in range(64)]) >= 0.99, scale
    for qi in range(64):  # every id both routes return carries the same exact distance
        e = {int(i): float(x) for i, x in zip(ei[qi], ed[qi])}
        for i, x in zip(ai[qi], ad[qi]):
            if int(i) in e:
                assert e[int(i)] == float(x)


def test_bruteforce_dot_metric(zv, oracle):
    V = oracle.uniform_rows(5, 0, 3000, 64) - 0.5
    Q = oracle.uniform_rows(6, 0, 32, 64) - 0.5
    idx = zv.Index(metric=zv.METRIC_DOT)
    idx.add(V)
    ri, rd, _ = oracle.bruteforce_topk(V, Q, 10, metric=1)
    for mode in (zv.BF_F32, zv.BF_I8):
        ids, dist, _ = idx.search_bruteforce(Q, 10, mode)
        assert np.array_equal(ids, ri) and np.array_equal(dist, rd)


def test_bruteforce_limits(zv, oracle):
    idx = zv.Index()
    idx.add(oracle.uniform_rows(1, 0, 10, 8))
    with pytest.raises(zv.ZvdbError):
        idx.search_bruteforce(oracle.uniform_rows(2, 0, 1, 8), 241)       # k > k' - margin (k' = 256: the stream routes' widest lists)
    wide = zv.Index()
    wide.add(oracle.uniform_rows(1, 0, 10, 200))
    with pytest.raises(zv.ZvdbError):
        wide.search_bruteforce(oracle.uniform_rows(2, 0, 1, 200), 113)    # fp32 rows beyond 128 dimensions: the staged kernel, k' = 128
    ids, dist, cnt = idx.search_bruteforce(oracle.uniform_rows(2, 0, 1, 8), 113)  # (fp32, dim <= 128: served; k > N returns N)
    ri, rd, _ = oracle.bruteforce_topk(oracle.uniform_rows(1, 0, 10, 8), oracle.uniform_rows(2, 0, 1, 8), 10)
    assert cnt[0] == 10 and np.array_equal(ids[0, :10], ri[0]) and np.array_equal(dist[0, :10], rd[0])
    with pytest.raises(zv.ZvdbError):
        idx.search_bruteforce(oracle.uniform_rows(2, 0, 1, 8), 5, zv.BF_BF16)  # no mirror
    empty = zv.Index(dim=8)
    ids, dist, cnt = empty.search_bruteforce(oracle.uniform_rows(2, 0, 2, 8), 5)
    assert np.all(cnt == 0)


@pytest.mark.parametrize("metric", ["l2", "dot", "cosine"])
def test_bf16_rows_storage_semantics(zv, oracle, metric):
    """ZVDB_FLAG_BF16_ROWS: elements are rounded to bf16 (RNE) when added, after the cosine
    normalisation; everything downstream is the exact fp32 arithmetic over the rounded values -- bit-exact against the
    oracle fed with oracle.bf16_round(rows), on every screen (the bf16 mirror of such rows is lossless)."""
    m = {"l2": zv.METRIC_L2SQ, "dot": zv.METRIC_DOT, "cosine": zv.METRIC_COSINE}[metric]
    n, d, nq, k = 6000, 96, 40, 10
    V = oracle.uniform_rows(21, 0, n, d) - (0.0 if metric == "l2" else 0.5)
    Q = oracle.uniform_rows(22, 0, nq, d) - (0.0 if metric == "l2" else 0.5)
    idx = zv.Index(metric=m, flags=zv.FLAG_BF16_MIRROR | zv.FLAG_BF16_ROWS)
    idx.add(V[:4000])
    idx.add(V[4000:])  # incremental adds round too
    stored = np.stack([idx.get_vector(i) for i in range(n)])
    assert not np.any(stored.view(np.uint32) & 0xFFFF)  # bf16-representable
    if metric == "cosine":
        plain = zv.Index(metric=m)
        plain.add(V)
        want = oracle.bf16_round(np.stack([plain.get_vector(i) for i in range(n)]))  # normalise first, then round
    else:
        want = oracle.bf16_round(V)
    assert np.array_equal(stored, want)
    ri, rd, _ = oracle.bruteforce_topk(want, Q, k, metric={"l2": 0, "dot": 1, "cosine": 1}[metric])
    for mode in (zv.BF_F32, zv.BF_BF16, zv.BF_I8):
        ids, dist, cnt = idx.search_bruteforce(Q, k, mode)
        if metric == "cosine":  # (queries are normalised by the index: compare the neighbours it ranks by -q.v / |q|)
            if mode == zv.BF_I8:  # 8-bit screen under an inner-product metric: near-exact, not guaranteed
                assert np.mean([len(set(ids[i]) & set(ri[i])) / k for i in range(nq)]) >= 0.99
            else:
                assert np.array_equal(ids, ri), mode
        else:
            assert np.array_equal(ids, ri) and np.array_equal(dist, rd), mode
    ai, ad, _ = idx.search_bruteforce(Q, k, zv.BF_FP4)
    assert np.mean([len(set(ai[i]) & set(ri[i])) / k for i in range(nq)]) >= 0.95


def test_fp4_screen_survivors_knob(zv, oracle):
    """zvdb_hip_index_set_screen_survivors: more fp4 survivors re-ranked -> recall can only grow; tight clusters far apart
    (neighbours differ by less than the 4-bit step resolves) are the case it is for.  Returned distances stay exact."""
    rng = np.random.default_rng(7)
    n, d, nq, k = 200_000, 64, 128, 10
    C = rng.standard_normal((32, d), dtype=np.float32) * 3
    V = (C[rng.integers(0, 32, n)] + rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    Q = (C[rng.integers(0, 32, nq)] + rng.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
    idx = zv.Index()
    idx.add(V)
    ei, ed, _ = idx.search_bruteforce(Q, k, zv.BF_I8)
    recs = []
    for sv in (0, 128, 256):
        idx.set_screen_survivors(sv)
        ai, ad, ac = idx.search_bruteforce(Q, k, zv.BF_FP4)
        assert np.all(ac == k) and np.all(np.diff(ad, axis=1) >= 0)
        recs.append(np.mean([len(set(ai[i]) & set(ei[i])) / k for i in range(nq)]))
        for qi in (0, nq - 1):
            assert oracle.distance(Q[qi], V[int(ai[qi, 0])]) == ad[qi, 0]
    assert recs[0] <= recs[1] + 0.01 and recs[1] <= recs[2] + 0.01 and recs[2] >= 0.9, recs
    with pytest.raises(zv.ZvdbError):
        idx.set_screen_survivors(100)


@pytest.mark.parametrize("n,d,nq,k", [(70_000, 1, 128, 1), (70_000, 2, 128, 10), (200_000, 3, 64, 10), (300_000, 6, 64, 10)])
def test_exact_screens_on_dense_low_dimensional_rows(zv, oracle, n, d, nq, k):
    """Many rows in a handful of dimensions sit inside the rounding band of the screens' |v|^2 - 2 q.v form (70k points on
    a line are 1e-5 apart): up to 4 dimensions every screen is evaluated in fp32, up to 8 the screens keep 128 survivors.
    The exact routes stay exact, bit for bit (found by tests/manual/fuzz_bruteforce.py)."""
    V = oracle.uniform_rows(1000 + n + d, 0, n, d)
    Q = oracle.uniform_rows(5000 + n + d, 0, nq, d)
    idx = _index(zv, V)
    ri, rd, _ = oracle.bruteforce_topk(V, Q, k)
    for mode in (zv.BF_F32, zv.BF_BF16, zv.BF_I8):
        ids, dist, _ = idx.search_bruteforce(Q, k, mode)
        assert np.array_equal(ids, ri) and np.array_equal(dist, rd), mode


def test_bf16_stream_results_do_not_depend_on_the_adaptive_xcd_split(zv, oracle):
    """The bf16 stream scan re-splits its rows over the 8 XCDs from the workgroup durations of earlier launches (bh_balance): every
    launch may run with a different partition of the rows into streams.  The results may not move: 30 launches of one batch, each
    synchronised so that the next one sees fresh durations, return the first launch's ids and distances, which are the int8 route's
    and (sampled) the oracle's exact top-k."""
    n, d, nq, k = 1_000_000, 64, 1024, 10
    idx = zv.Index(flags=zv.FLAG_BF16_MIRROR)
    idx.add_synthetic(n, d, 0x5EED)
    Q = oracle.uniform_rows(0xC0FFEE, 0, nq, d)
    first = idx.search_bruteforce(Q, k, zv.BF_BF16)
    for _ in range(30):
        again = idx.search_bruteforce(Q, k, zv.BF_BF16)
        assert all(np.array_equal(a, b) for a, b in zip(first, again))
    i8 = idx.search_bruteforce(Q, k, zv.BF_I8)
    assert np.array_equal(first[0], i8[0]) and np.array_equal(first[1], i8[1])
    V = oracle.uniform_rows(0x5EED, 0, n, d)
    sel = [0, 511, 1023]
    ri, rd, _ = oracle.bruteforce_topk(V, Q[sel], k)
    assert np.array_equal(first[0][sel], ri) and np.array_equal(first[1][sel], rd)


def test_bruteforce_dev_from_two_streams_with_two_workspaces(zv, oracle):
    """zvdb_hip_search_bruteforce_dev keeps its per-call state in the CALLER's workspace: two batches launched on two streams of one
    index, each with a workspace of its own, return what the same calls return one after the other -- on the bf16 stream route
    (guessed thresholds, adaptive split: the index-side state they share is host bookkeeping under the handle's mutex) and on the
    int8 route."""
    import torch
    n, d, nq, k = 600_000, 64, 512, 10
    idx = zv.Index(flags=zv.FLAG_BF16_MIRROR)
    idx.add_synthetic(n, d, 0x5EED)
    dev = torch.device("cuda", 0)
    qa = torch.from_numpy(oracle.uniform_rows(0xC0FFEE, 0, nq, d)).to(dev)
    qb = torch.from_numpy(oracle.uniform_rows(0xC0FFEE, nq, nq, d)).to(dev)
    for mode in (zv.BF_BF16, zv.BF_I8):
        wsb = idx.bruteforce_workspace(nq, k, mode)
        ws = [torch.empty(wsb, dtype=torch.uint8, device=dev) for _ in range(2)]

        def run(q, w, stream):
            ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
            cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
            idx.search_bruteforce_dev(q.data_ptr(), nq, d, k, mode, 0, w.data_ptr(), wsb, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), stream.cuda_stream)
            return ids, dd, cnt

        s0 = torch.cuda.current_stream()
        ra = run(qa, ws[0], s0); rb = run(qb, ws[1], s0)
        torch.cuda.synchronize()
        want = [t.cpu().numpy().copy() for t in ra + rb]
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        s1.wait_stream(s0); s2.wait_stream(s0)
        for _ in range(6):
            ga = run(qa, ws[0], s1); gb = run(qb, ws[1], s2)
            torch.cuda.synchronize()
            got = [t.cpu().numpy() for t in ga + gb]
            assert all(np.array_equal(a, b) for a, b in zip(want, got)), mode
    V = oracle.uniform_rows(0x5EED, 0, n, d)
    ri, rd, _ = oracle.bruteforce_topk(V, qa[:3].cpu().numpy(), k)
    assert np.array_equal(want[0][:3].astype(np.uint64), ri) and np.array_equal(want[1][:3], rd)
