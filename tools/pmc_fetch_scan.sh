# usage (on the GPU box): bash tools/pmc_fetch_scan.sh <tag> [ENV=VALUE ...]  -- FETCH_SIZE per launch of the bf16 stream scan (one PMC pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmcf_$TAG -o p --output-format csv -- python3 $R/bench.py --mode bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-paths > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/pmcf_$TAG/**/*counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == "FETCH_SIZE" and "k_bf16_stream<48, 1, false, 64, 0>" in r["Kernel_Name"]]
print("$TAG", "$*", "scan launches", len(v), "2 x FETCH_SIZE per launch (GB): mean %.2f min %.2f max %.2f" % (2 * sum(v) / len(v) * 1024 / 1e9, 2 * min(v) * 1024 / 1e9, 2 * max(v) * 1024 / 1e9))
PY
