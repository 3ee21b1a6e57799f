#!/bin/bash
# SQ counters of the headline scan kernel per BUILD of libzvdb_hip.so on one box (one --pmc pass each, --kernel-trace only):
#   bash tools/pmc_ab.sh "tree ab_x.so" [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LIBS=$1; shift
cp $R/zvdb_amd/libzvdb_hip.so /tmp/ab_tree.so
trap 'cp /tmp/ab_tree.so $R/zvdb_amd/libzvdb_hip.so' EXIT
for v in $LIBS; do
  if [ "$v" = tree ]; then cp /tmp/ab_tree.so $R/zvdb_amd/libzvdb_hip.so; else cp "$R/$v" $R/zvdb_amd/libzvdb_hip.so; fi
  rm -rf /tmp/pmcab
  rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    -d /tmp/pmcab -o p --output-format csv -- python3 $R/bench.py --mode bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-paths "$@" > /dev/null 2> /tmp/pmcab_err.txt || { tail -5 /tmp/pmcab_err.txt; exit 1; }
  python3 - "$v" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pmcab/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if "k_bf16_stream<48, 1, false" not in r["Kernel_Name"] or ", 1>(" in r["Kernel_Name"]:
        continue
    a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
o = {k: v / max(n, 1) for k, (v, n) in acc.items()}
n = max(n for _, n in acc.values())
print(f"{sys.argv[1]:>20} (mean of {n} scan launches): " + "  ".join(f"{k}={v:.5g}" for k, v in sorted(o.items())), flush=True)
if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "SQ_BUSY_CYCLES" in o:
    print(f"{sys.argv[1]:>20}   MFMA busy / SQ busy cycles (x4 SIMDs) = {o['SQ_VALU_MFMA_BUSY_CYCLES'] / o['SQ_BUSY_CYCLES'] / 4:.3f};  wave cycles waiting on an instruction = {o['SQ_WAIT_INST_ANY'] / o['SQ_WAVE_CYCLES']:.3f} (on LDS {o['SQ_WAIT_INST_LDS'] / o['SQ_WAVE_CYCLES']:.3f});  GRBM_GUI_ACTIVE / 8 = {o['GRBM_GUI_ACTIVE'] / 8:.5g} cycles", flush=True)
PY
done
