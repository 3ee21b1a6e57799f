cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# the harness is built here, from its source (no binary in the tree)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/bf16_regq_bench.hip -o $R/tools/bf16_regq_bench
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_regq -o p --output-format csv -- $R/tools/bf16_regq_bench 10000000 > $R/gpurun_out/pmc_regq.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/pmc_regq/**/*counter_collection.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "FETCH_SIZE" and "k_regq" in r["Kernel_Name"]:
        print(r["Kernel_Name"][:40], r["Grid_Size"], float(r["Counter_Value"]) * 2 / 1e6, "GB (x2 corrected)")
PY
