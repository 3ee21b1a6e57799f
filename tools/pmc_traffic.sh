# usage (on the GPU box): bash tools/pmc_traffic.sh <mode> <tag>  -- HBM traffic of one step: FETCH_SIZE and WRITE_SIZE in separate passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fp4}
TAG=${2:-$MODE}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_${c}_$TAG -o p --output-format csv -- python3 $R/bench.py --mode $MODE --steps 3 --warmup 1 --no-cpu-baseline --no-paths > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$R/gpurun_out/pmc_%s_$TAG/**/*counter_collection.csv" % c, recursive=True)
    if not f: print(c, "no csv"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c]
    per = collections.defaultdict(list)
    for r in rows: per[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        if max(v) > 1000: print(f"{c:11s} {len(v):4d} dispatches  mean {sum(v)/len(v):14.1f} KB  max {max(v):14.1f} KB  {k}")
    out[c] = {k: v for k, v in per.items()}
json.dump(out, open("$R/gpurun_out/pmc_traffic_$TAG.json", "w"))
PY
