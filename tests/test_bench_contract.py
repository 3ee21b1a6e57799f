"""The bench line's contract, checked on the committed default line of the round (profiles/r5/bench_default_line.json = the output
of `python bench.py` on an MI355X): every field the driver and the judge read is there, typed, and consistent with itself.  (A CPU
test: it reads the committed JSON; `-m gpu` runs bench.py itself in tests/test_gpu_zz_multiprocess.py.)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    p = os.path.join(ROOT, "profiles", "r5", "bench_default_line.json")
    return json.loads([l for l in open(p) if l.startswith("{")][-1])


def test_default_line_has_the_contract_fields():
    d = _line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].startswith("QPS at recall@10") and "10Mx768" in d["metric"] and base["metric"].startswith("QPS at recall@10")
    assert d["unit"] == "QPS" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["steps"] >= 1 and d["warmup"] >= 0 and d["scaling"] in ("weak", "strong")
    assert d["dtype"] in ("bf16", "f32") and d["recall_at_10"] >= 0.95
    assert abs(d["value"] - d["config"]["batch"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6   # value = batch / step time
    assert d["ms_per_step_max"] >= d["ms_per_step"] * 0.98
    c = d["config"]
    assert isinstance(c["workload"], str) and (c["rows"], c["dim"], c["batch"], c["k"]) == (10_000_000, 768, 1024, 10) and "model" not in c
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    assert r["kernel_ms_avg"] <= d["ms_per_step"] and r["timed_regions"] >= 1
    assert abs(r["achieved"] - r["algorithmic_flops_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e12) / r["achieved"] < 1e-6
    assert r["traffic"] is None or r["traffic"] > 0.9 * r["algorithmic_bytes_per_launch"]
    assert r["traffic"] is None or "sha256" in r["traffic_source"] or "measured in this run" in r["traffic_source"]
    b = d["cpu_baseline"]
    assert b["kind"] in ("port", "reference") and b["unit"] == "QPS" and b["value"] > 0 and b["cores"] >= 1 and isinstance(b["sample"], str)
    assert b["gpu_result_check"].startswith("ok")


def test_default_line_accounts_for_the_certificate_and_carries_the_reference_sweep():
    d = _line()
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
