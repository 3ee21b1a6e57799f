#!/bin/bash
# durations of every launch of the configs[1] scan kernel over 400 identical steps (rocprofv3 --kernel-trace of tools/c2_trace.py 400 [dim]): percentiles and the series
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ld
rocprofv3 --kernel-trace --output-format csv -d /tmp/ld -o s -- python3 $R/tools/c2_trace.py ${1:-400} ${2:-128} > /dev/null 2> /tmp/ld_err.txt || { tail -5 /tmp/ld_err.txt; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ld/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_f32_regq_" in r["Kernel_Name"] and not r["Kernel_Name"].split("(")[0].endswith("_1")]
s = sorted(d)
n = len(s)
print(f"{n} launches: min {s[0]:.1f}  p10 {s[n // 10]:.1f}  median {s[n // 2]:.1f}  p90 {s[9 * n // 10]:.1f}  p99 {s[99 * n // 100]:.1f}  max {s[-1]:.1f}  mean {sum(s) / n:.1f} us")
print("series (us, every launch):")
for i in range(0, n, 20): print("  " + " ".join(f"{x:5.0f}" for x in d[i:i + 20]))
PY
