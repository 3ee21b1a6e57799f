#!/bin/bash
# round-4 profile collection (run from the repo root on the GPU box): headline step stats + HBM traffic passes, graph kernels
# (stats at 1M rows, FETCH_SIZE at 200k rows: rocprofv3's counter pass dies at 1M on this image) at search widths 1 and 4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
tools/prof_bench_r4.sh default > /dev/null
bash tools/pmc_traffic.sh bf16 r4 > gpurun_out/pmc_traffic_r4.txt 2>&1
for W in 1 4; do
  export ZVDB_PROFILE_WIDTH=$W
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_graph_w$W -o s -- python3 $R/tests/manual/graph_profile.py f32 bf16 > $R/gpurun_out/stats_graph_w$W.log 2>&1
  cp $(find $R/gpurun_out/stats_graph_w$W -name "*kernel_stats.csv" | head -1) $R/gpurun_out/stats_graph_w${W}_kernel_stats.csv
  rm -rf $R/gpurun_out/stats_graph_w$W
  for L in f32 bf16; do
    bash tools/prof_graph_pmc.sh $L > gpurun_out/pmc_graph_${L}_w$W.txt 2>&1
    cp gpurun_out/pmc_graph_fetch_$L.csv gpurun_out/pmc_graph_fetch_${L}_w$W.csv
    cp gpurun_out/graph_profile_counters_${L}_w$W.json gpurun_out/graph_profile_counters_200k_${L}_w$W.json 2>/dev/null
    rm -rf gpurun_out/pmc_graph_$L
  done
done
rm -rf gpurun_out/pmc_FETCH_SIZE_r4 gpurun_out/pmc_WRITE_SIZE_r4
grep -E "^ef128" gpurun_out/stats_graph_w*.log | cut -c1-400
cat gpurun_out/pmc_graph_*_w*.txt | grep -v amdgpu | tail -20
tail -12 gpurun_out/pmc_traffic_r4.txt
