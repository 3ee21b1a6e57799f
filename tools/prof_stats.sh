# usage (on the GPU box): bash tools/prof_stats.sh <mode> <tag>  -- rocprofv3 kernel stats of a short bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fp4}
TAG=${2:-$MODE}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$TAG -o s -- python3 $R/bench.py --mode $MODE --steps 10 --warmup 2 --no-cpu-baseline --no-paths > $R/gpurun_out/stats_$TAG.line 2>&1
f=$(find $R/gpurun_out/stats_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/stats_${TAG}_kernel_stats.csv
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/stats_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# the scan launches of the steady state, in launch order: print the last step's chain
ks = [(r["Kernel_Name"][:70], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size"])) for r in rows]
last = [i for i, k in enumerate(ks) if "k_rerank" in k[0]]
if len(last) >= 2:
    for k in ks[last[-2] + 1:last[-1] + 1]: print(f"{k[1]:10.1f} us  grid {k[2]:8d}  {k[0]}")
PY
head -8 $R/gpurun_out/stats_${TAG}_kernel_stats.csv | cut -c1-200
