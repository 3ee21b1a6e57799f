#!/bin/bash
# rocprofv3 kernel stats of the default bench step (run from the repo root on the GPU box); output under gpurun_out/prof_r4_<tag>/
# usage: tools/prof_bench_r4.sh <tag> [bench.py arguments]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-default}; shift
OUT=$R/gpurun_out/prof_r4_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --steps 12 --warmup 6 --no-paths --no-cpu-baseline "$@" > $OUT/line.json 2> $OUT/err.txt
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python3 $R/tools/trace_step.py $OUT/stats > $OUT/step_trace.txt 2>&1
head -22 $OUT/kernel_stats.csv | cut -c1-220
rm -rf $OUT/stats
