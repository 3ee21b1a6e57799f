#!/bin/bash
# A/B of several builds of libzvdb_hip.so on ONE box by the KERNEL's own duration: bench.py under rocprofv3 --kernel-trace --stats per build,
# the average / min duration of every k_bf16_stream symbol printed (the non-gated scan is the one that counts; timing-only ablations
# return wrong results, so step times are confounded by repairs -- the scan kernel's duration is not).
#   bash tools/ab_kernel_stats.sh "tree ab_x.so" [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LIBS=$1; shift
cp $R/zvdb_amd/libzvdb_hip.so /tmp/ab_tree.so
trap 'cp /tmp/ab_tree.so $R/zvdb_amd/libzvdb_hip.so' EXIT
for v in $LIBS; do
  if [ "$v" = tree ]; then cp /tmp/ab_tree.so $R/zvdb_amd/libzvdb_hip.so; else cp "$R/$v" $R/zvdb_amd/libzvdb_hip.so; fi
  rm -rf /tmp/abks
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abks -o s -- python3 $R/bench.py --mode bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-paths "$@" > /tmp/abks_line.json 2> /tmp/abks_err.txt || { tail -5 /tmp/abks_err.txt; exit 1; }
  python3 - "$v" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/abks/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_bf16_stream<48" in r["Name"] and int(r["Calls"]) >= 10:
        nm = r["Name"].split("k_bf16_stream")[1].split("(")[0]
        print(f"{sys.argv[1]:>24} k_bf16_stream{nm}: calls {r['Calls']} avg {float(r['AverageNs']) / 1e6:.4f} ms min {float(r['MinNs']) / 1e6:.4f} ms", flush=True)
PY
done
