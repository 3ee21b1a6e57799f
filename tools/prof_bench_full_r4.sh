#!/bin/bash
# rocprofv3 kernel stats of the WHOLE default bench line (headline + every leg under `paths`), for the legs' roofline objects
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r4_full
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/line.json 2> $OUT/err.txt
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
python3 - <<PY
import csv, json
d = json.load(open("$OUT/line.json"))
print("headline", round(d["value"]), d["roofline"]["kernel_ms_avg"])
for k, v in d["paths"].items(): print(k, round(v["value"]), v["roofline"].get("kernel_ms_avg"))
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    if any(t in r["Name"] for t in ("k_search_ef_reg", "k_bf16_stream<48, 1, false", "k_bf_scan5<0", "k_descend", "k_i8_stream", "k_fp4_stream")):
        print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg")
PY
