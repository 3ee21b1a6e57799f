#!/bin/bash
# A/B of several builds of libzvdb_hip.so on ONE box by the launches of a configs[1] step (rocprofv3 --kernel-trace of tools/c2_trace.py 400: the last step's
# timeline with its gaps, and the mean step span of the last 100 steps).     bash tools/ab_c2_steps.sh "tree ab_x.so"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LIBS=$1
cp $R/zvdb_amd/libzvdb_hip.so /tmp/ab_tree.so
trap 'cp /tmp/ab_tree.so $R/zvdb_amd/libzvdb_hip.so' EXIT
for v in $LIBS; do
  if [ "$v" = tree ]; then cp /tmp/ab_tree.so $R/zvdb_amd/libzvdb_hip.so; else cp "$R/$v" $R/zvdb_amd/libzvdb_hip.so; fi
  rm -rf /tmp/abc2
  rocprofv3 --kernel-trace --output-format csv -d /tmp/abc2 -o s -- python3 $R/tools/c2_trace.py 400 > /dev/null 2> /tmp/abc2_err.txt || { tail -5 /tmp/abc2_err.txt; exit 1; }
  echo "== $v"
  python3 $R/tools/trace_step.py /tmp/abc2
  python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/abc2/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [int(r["End_Timestamp"]) for r in rows if "k_rerank" in r["Kernel_Name"]]
d = [(b - a) / 1e3 for a, b in zip(ends[-101:-1], ends[-100:])]
print(f"mean step period of the last 100 steps: {sum(d) / len(d):.1f} us")
PY
done
