#!/bin/bash
# rocprofv3 kernel stats of the large-beam searches (tools/ef_large_probe.py, the library's own choice of waves per query, width 4)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r4_ef_large
mkdir -p $OUT
export ZVDB_PROBE_LEGS=rank12_bulk_bf16rows ZVDB_PROBE_COMBOS=auto
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/tools/ef_large_probe.py 1000000 768 256 512 1024 > $OUT/out.txt 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv; rm -rf $OUT/stats
grep -v amdgpu $OUT/out.txt | cut -c1-400; grep "k_search_ef_reg\|k_search_ef<\|k_descend" $OUT/kernel_stats.csv | cut -c1-90,300-420
