cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_calib -o p --output-format csv -- $R/tools/mfma_peak_bench > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_calib/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
by = collections.defaultdict(dict)
for r in rows: by[(r["Dispatch_Id"], r["Kernel_Name"][:40], r["Grid_Size"])][r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in list(by.items())[:16]: print(k, {a: f"{b:.4g}" for a, b in v.items()})
PY
