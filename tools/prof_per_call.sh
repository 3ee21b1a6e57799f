#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r4_per_call
mkdir -p $OUT
python3 $R/tools/per_call_probe.py > $OUT/wall.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/tools/per_call_probe.py > $OUT/wall_profiled.txt 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv; rm -rf $OUT/stats
cat $OUT/wall.txt | grep -v amdgpu; head -8 $OUT/kernel_stats.csv | cut -c1-60,280-400
