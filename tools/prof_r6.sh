#!/bin/bash
# Round-6 profile set of the default bench command, on ONE lease (run from the repo root on the GPU box):
#   1. the default line itself                                   -> gpurun_out/r6p/bench_default_line.json
#   2. rocprofv3 --kernel-trace --stats of the same command      -> kernel_stats.csv, step trace of the headline (passes.txt)
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes   -> traffic_10Mx768.json, stamped with the scan kernel's symbol and
#      the sha256 of its source file (bench.py refuses the file when the stamp does not match the tree)
# Copy gpurun_out/r6p/* to profiles/r6/ afterwards.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6p
mkdir -p $OUT
python3 $R/bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default_line.err || { tail -5 $OUT/bench_default_line.err; exit 1; }
cp $R/bench_full.json $OUT/bench_full.json
echo "default line done ($(wc -c < $OUT/bench_default_line.json) bytes on stdout)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_profiled_line.json 2> $OUT/prof.err || { tail -5 $OUT/prof.err; exit 1; }
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_default_full_line_kernel_stats.csv
rm -rf $OUT/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --mode bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-paths > $OUT/bench_10Mx768_bf16_stream_profiled_line.json 2> $OUT/prof2.err || { tail -5 $OUT/prof2.err; exit 1; }
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_10Mx768_bf16_stream_kernel_stats.csv
( echo "# launches of one headline step in order (rocprofv3 --kernel-trace of python3 bench.py --mode bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-paths)"; python3 $R/tools/trace_step.py $OUT/stats ) > $OUT/bench_10Mx768_bf16_stream_passes.txt 2>&1
rm -rf $OUT/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/tools/c2_trace.py 400 > /dev/null 2> $OUT/prof3.err || { tail -5 $OUT/prof3.err; exit 1; }
( echo "# launches of one configs[1] step in order (1M x 128 fp32, batch 256, fp32 screen; rocprofv3 --kernel-trace --stats of tools/c2_trace.py 400: the LAST of 400 steps, clocks warm)"; python3 $R/tools/trace_step.py $OUT/stats; echo "# kernel stats of the same run (all 400 steps): name, calls, average us, min us, max us"; python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("k_f32_regq", "k_rq_guess", "k_bf_merge_sel", "k_rerank", "k_exact_scan")):
        nm = r["Name"].replace("zvdb::", "").replace("(anonymous namespace)::", "").split("(")[0]
        print(f"{nm:48s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e3:9.1f} {float(r['MinNs']) / 1e3:9.1f} {float(r['MaxNs']) / 1e3:9.1f}")
PY
) > $OUT/c2_1Mx128_f32_passes.txt 2>&1
rm -rf $OUT/stats
echo "kernel stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o p --output-format csv -- python3 $R/bench.py --mode bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-paths > /dev/null 2> $OUT/pmc_$c.err || { tail -5 $OUT/pmc_$c.err; exit 1; }
done
python3 - <<PY
import csv, glob, hashlib, json
R, OUT = "$R", "$OUT"
KEY = "k_bf16_stream<48, 1, false"
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{OUT}/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and KEY in r["Kernel_Name"] and ", 1>(" not in r["Kernel_Name"]]  # (", 1>": the gated no-op launch)
    per[c] = [float(r["Counter_Value"]) for r in rows]
    per[c] = [v for v in per[c] if v > 0.01 * max(per[c])]   # (the gated repair launch shares the symbol's prefix: a no-op that moves nothing)
    name = rows[0]["Kernel_Name"].split("(")[0]
fb = 2.0 * sum(per["FETCH_SIZE"]) / len(per["FETCH_SIZE"]) * 1024
wb = sum(per["WRITE_SIZE"]) / len(per["WRITE_SIZE"]) * 1024
alg = 1.0e7 * 768 * 2 + 4.0e7
sha = hashlib.sha256(open(f"{R}/zvdb_amd/csrc/bf_bf16_stream.hip", "rb").read()).hexdigest()
out = {"bf16": {"rows": 10000000, "dim": 768, "batch": 1024, "kernel": name, "kernel_source": "zvdb_amd/csrc/bf_bf16_stream.hip", "kernel_source_sha256": sha,
                "fetch_kb_per_scan_launch": per["FETCH_SIZE"], "write_kb_per_scan_launch": per["WRITE_SIZE"], "fetch_bytes_corrected": fb, "write_bytes": wb,
                "hbm_bytes_per_step": fb + wb, "over_algorithmic": (fb + wb) / alg,
                "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (--kernel-trace only) of python3 bench.py --mode bf16 --steps 3 --warmup 1 "
                        "--no-cpu-baseline --no-paths (tools/prof_r6.sh), the scan dispatches only; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 16-B/lane correction); "
                        "algorithmic 15.36 GB + 0.04 GB of start values"}}
json.dump(out, open(f"{OUT}/traffic_10Mx768.json", "w"), indent=1)
print("hbm bytes per scan %.4g = %.3fx of the algorithmic %.4g (%d / %d dispatches)" % (fb + wb, (fb + wb) / alg, alg, len(per["FETCH_SIZE"]), len(per["WRITE_SIZE"])))
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_full.json").read())
print("headline", round(d["value"]), "QPS frac", round(d["roofline"]["frac"], 4), "step max", d.get("ms_per_step_max"), "settle", {k: d["settle"][k] for k in ("settle_ms_total", "repair_batches", "ms_per_repair_batch")})
for k, v in d["paths"].items(): print(k, round(v["value"]), round(v["roofline"]["frac"], 3) if "roofline" in v else "")
PY
