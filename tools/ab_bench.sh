#!/bin/bash
# A/B/... of several builds of libzvdb_hip.so on ONE box (boxes of the pool differ by 5-7 %): runs bench.py in turn with
# every library listed (round-robin, ROUNDS times); "tree" = zvdb_amd/libzvdb_hip.so as built.
#   gpurun -- 'bash tools/ab_bench.sh 3 "tree ab_base.so" --mode bf16 --no-cpu-baseline --no-paths'
set -e
ROUNDS=$1; LIBS=$2; shift 2
cp zvdb_amd/libzvdb_hip.so /tmp/ab_tree.so
trap 'cp /tmp/ab_tree.so zvdb_amd/libzvdb_hip.so' EXIT
for r in $(seq 1 $ROUNDS); do
  for v in $LIBS; do
    if [ "$v" = tree ]; then cp /tmp/ab_tree.so zvdb_amd/libzvdb_hip.so; else cp "$v" zvdb_amd/libzvdb_hip.so; fi
    python bench.py --steps 20 --warmup 5 "$@" > /tmp/ab_line.json 2> /tmp/ab_err.txt || { tail -5 /tmp/ab_err.txt; exit 1; }
    python - "$v" "$r" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[1]:>24} round {sys.argv[2]}: {d['value']:.0f} QPS  step {d['ms_per_step']:.3f} ms  scan {r['kernel_ms_avg']:.3f} ms  frac {r['frac']:.4f}  recall {d.get('recall_at_10')}", flush=True)
PY
  done
done
