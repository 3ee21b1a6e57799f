#!/bin/bash
# per-launch timeline of one BASELINE configs[1] step (1M x 128 fp32, batch 256, fp32 screen): tools/c2_trace.py under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6c/prof_c2
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/tools/c2_trace.py ${C2_ITERS:-6} > $OUT/out.txt 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 $R/tools/trace_step.py $OUT/stats | tee $OUT/step_trace.txt
rm -rf $OUT/stats
