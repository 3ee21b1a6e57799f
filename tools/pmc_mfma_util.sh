# usage (on the GPU box): bash tools/pmc_mfma_util.sh  -- SQ counters of the headline scan (k_bf16_stream<48,1,false,64,0>): MFMA busy against
# the kernel's cycles, wave-state split, instruction mix.  Two --pmc passes, --kernel-trace only (MI355X_MICROARCH.md: 8 SQ slots per pass).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_mfma_$i -o p --output-format csv -- python3 $R/bench.py --mode bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-paths > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
out = {}
for d in sorted(glob.glob("$R/gpurun_out/pmc_mfma_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if "k_bf16_stream<48, 1, false, 64, 0>" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in acc.items():
        out[k] = v / max(n, 1)
        print(f"{k:28s} {v / max(n,1):.6g}  (mean over {n} scan launches)")
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "SQ_BUSY_CU_CYCLES" in out:
    print("MFMA busy / CU busy cycles: %.3f" % (out["SQ_VALU_MFMA_BUSY_CYCLES"] / out["SQ_BUSY_CU_CYCLES"]))
if "GRBM_GUI_ACTIVE" in out:
    print("GRBM_GUI_ACTIVE / 8 XCDs = %.4g cycles per launch" % (out["GRBM_GUI_ACTIVE"] / 8))
if "SQ_INSTS_MFMA" in out:
    print("MFMA instructions per launch: %.6g (expected 2 x 1024 x 10^7 x 768 / 16384 = %.6g wave-instructions of 16x16x32)" % (out["SQ_INSTS_MFMA"], 2 * 1024 * 1e7 * 768 / 2 / 8192))
PY
