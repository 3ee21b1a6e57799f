"""The bench line's contract, checked (a) on the committed default line of the round -- profiles/r6/bench_default_line.json = the LAST
stdout line of `python bench.py` on an MI355X, profiles/r6/bench_full.json = the full record it wrote beside it -- and (b) on bench.py's
own compaction code applied to that full record: every field the driver and the judge read is there, typed, small enough for a record
that keeps only the tail of stdout (round 5's 20-KB line was lost to that), and consistent with itself.  (CPU tests; `-m gpu` runs
bench.py itself and holds what it prints to the same rules: tests/test_gpu_zz_multiprocess.py:_bench.)"""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE_LIMIT = 4096


def _line_text():
    p = os.path.join(ROOT, "profiles", "r6", "bench_default_line.json")
    return [l for l in open(p).read().splitlines() if l.startswith("{")][-1]


def _full():
    return json.load(open(os.path.join(ROOT, "profiles", "r6", "bench_full.json")))


def _check_contract(d):
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].startswith("QPS at recall@10") and "10Mx768" in d["metric"] and base["metric"].startswith("QPS at recall@10")
    assert d["unit"] == "QPS" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["steps"] >= 1 and d["warmup"] >= 0 and d["scaling"] in ("weak", "strong")
    assert d["dtype"] in ("bf16", "f32") and d["recall_at_10"] >= 0.95
    assert abs(d["value"] - d["config"]["batch"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3   # value = batch / step time
    assert d["ms_per_step_max"] >= d["ms_per_step"] * 0.98
    c = d["config"]
    assert isinstance(c["workload"], str) and len(c["workload"]) <= 160 and "model" not in c
    assert (c["rows"], c["dim"], c["batch"], c["k"]) == (10_000_000, 768, 1024, 10) and c["settle_batches_before_warmup"] == 32
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0.3 < r["frac"] < 1.0
    assert r["kernel_ms_avg"] <= d["ms_per_step"] and r["timed_regions"] >= 1
    assert abs(r["achieved"] - r["algorithmic_flops_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e12) / r["achieved"] < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0.9 * r["algorithmic_bytes_per_launch"]
    assert "(" not in r["kernel"] and len(r["kernel"]) < 40      # the symbol only
    b = d["cpu_baseline"]
    assert b["kind"] in ("port", "reference") and b["unit"] == "QPS" and b["value"] > 0 and b["cores"] >= 1 and isinstance(b["sample"], str)
    assert b["gpu_result_check"].startswith("ok")
    h = b["reference_harness"]
    assert h["bit_exact"] is True and h["cpu_search_qps"] > 0 and h["gpu_search_qps_batched"] > 0 and h["gpu_search_per_call_us"] > 0
    assert h["sweep"]["all_bit_exact"] is True and h["sweep"]["cells"] >= 16
    assert d["full_record"] == "bench_full.json"
    for name in ("graph_ef128_1Mx768_rank12_bf16rows", "c2_1Mx128_b256_f32"):
        assert 0.0 < d["paths"][name]["frac"] < 1.0


def test_committed_line_is_small_and_has_the_contract_fields():
    text = _line_text()
    assert len(text) < LINE_LIMIT, len(text)
    _check_contract(json.loads(text))


def test_compact_line_of_the_committed_full_record():
    """bench.py's own compaction (the function the live run prints with) applied to the committed full record."""
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = _full()
    text = bench.compact_line(full, "bench_full.json")
    assert len(text) < LINE_LIMIT and bench.LINE_LIMIT == LINE_LIMIT
    d = json.loads(text)
    _check_contract(d)
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-6
    # an N > 1 record keeps its shape: cpu_baseline with a null value, `collective` numbers
    multi = dict(full, n_gpus=4, cpu_baseline={"value": None, "unit": "QPS", "cores": 0, "kind": "port", "sample": "not timed at N > 1"},
                 collective={"ranks": 4, "ranks_reporting": 4, "distinct_devices": 4, "all_gather_us": 31.0, "bytes_per_rank": 127232,
                             "source": "zvdb_hip_comm_query (ncclCommUserRank / ncclCommCount) + HIP events around ncclAllGather inside libzvdb_hip"})
    m = json.loads(bench.compact_line(multi, "bench_full.json"))
    assert m["cpu_baseline"]["value"] is None and m["collective"]["ranks"] == 4 and m["collective"]["via"] == "rccl_in_lib"


def test_full_record_accounts_for_the_certificate_and_carries_the_reference_sweep():
    d = _full()
    s = d["settle"]
    assert s["batches"] == 32 and len(s["repaired_per_batch"]) == 32 and s["settle_ms_total"] > 0
    assert s["repair_batches"] == sum(1 for r in s["repaired_per_batch"] if r > 0)
    assert s["repaired_queries_in_warmup_and_timed_steps"] == 0          # the timed region ran on the settled index
    assert d["recall_batches"]["batches"] >= 4 and min(d["recall_batches"]["per_batch"]) >= 0.95
    sw = d["cpu_baseline"]["reference_harness"]["sweep"]
    cells = [c for k, c in sw["cells"].items() if not k.endswith("_insert")]
    assert len(cells) == 16 and sw["all_bit_exact"] is True
    assert sorted({c["dim"] for c in cells}) == [128, 512, 768, 1024] and sorted({c["k"] for c in cells}) == [10, 25, 50, 100]
    assert all(c["queries"] <= 2000 and c["bit_exact_first_100_batched"] and c["bit_exact_first_100_per_call"] for c in cells)
    for name in ("graph_ef128_1Mx768_rank12_bf16rows", "c2_1Mx128_b256_f32"):
        assert 0.0 < d["paths"][name]["roofline"]["frac"] < 1.0
    assert "k_f32_regq" in d["paths"]["c2_1Mx128_b256_f32"]["roofline"]["kernel"]
