# usage (on the GPU box): bash tools/prof_stats.sh <mode> <tag>  -- rocprofv3 kernel stats of a short bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fp4}
TAG=${2:-$MODE}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$TAG -o s -- python3 $R/bench.py --mode $MODE --steps 10 --warmup 2 --no-cpu-baseline --no-paths > $R/gpurun_out/stats_$TAG.line 2>&1
f=$(find $R/gpurun_out/stats_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/stats_${TAG}_kernel_stats.csv
python3 $R/tools/trace_step.py $R/gpurun_out/stats_$TAG
head -8 $R/gpurun_out/stats_${TAG}_kernel_stats.csv | cut -c1-200
