#!/bin/bash
# SQ counters + clock of the configs[1] scan kernel (k_f32_regq_16_64_0; [dim] picks another geometry) -- one --pmc pass, --kernel-trace only:
#   bash tools/pmc_c2.sh [dim]          prints MFMA-busy / SQ-busy, wave cycles waiting, and the clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
DIM=${1:-128}
rm -rf /tmp/pmcc2
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU \
  -d /tmp/pmcc2 -o p --output-format csv -- python3 $R/tools/c2_trace.py 200 $DIM > /dev/null 2> /tmp/pmcc2_err.txt || { tail -5 /tmp/pmcc2_err.txt; exit 1; }
python3 - "$DIM" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pmcc2/**/*counter_collection.csv", recursive=True)[0]
t = glob.glob("/tmp/pmcc2/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if "k_f32_regq_" not in r["Kernel_Name"] or r["Kernel_Name"].split("(")[0].endswith("_1"):
        continue
    a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
o = {k: v / max(n, 1) for k, (v, n) in acc.items()}
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(t)) if "k_f32_regq_" in r["Kernel_Name"] and not r["Kernel_Name"].split("(")[0].endswith("_1")]
d = d[len(d) // 2:]   # (warm half)
dur = sum(d) / len(d)
fl = 2.0 * 256 * 1e6 * int(sys.argv[1])
print(f"dim {sys.argv[1]}: scan kernel under --pmc {dur / 1e3:.1f} us average ({len(d)} launches) = {fl / (dur * 1e-9) / 157.3e12:.3f} of the fp32-MFMA peak")
print("  " + "  ".join(f"{k}={v:.5g}" for k, v in sorted(o.items())))
clk = o["GRBM_GUI_ACTIVE"] / 8 / (dur * 1e-9) / 1e9
print(f"  clock = GRBM_GUI_ACTIVE / 8 XCDs / duration = {clk:.3f} GHz;  MFMA busy / SQ busy cycles (x4 SIMDs) = {o['SQ_VALU_MFMA_BUSY_CYCLES'] / o['SQ_BUSY_CYCLES'] / 4:.3f};  "
      f"wave cycles waiting on an instruction = {o['SQ_WAIT_INST_ANY'] / o['SQ_WAVE_CYCLES']:.3f}")
print(f"  => fraction of peak = MFMA busy x clock / 2.4 GHz = {o['SQ_VALU_MFMA_BUSY_CYCLES'] / o['SQ_BUSY_CYCLES'] / 4 * clk / 2.4:.3f}")
PY
