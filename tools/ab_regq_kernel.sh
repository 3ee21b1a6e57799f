#!/bin/bash
# A/B of several builds of libzvdb_hip.so on ONE box by the fp32 register-query scan KERNEL's own duration (rocprofv3 --kernel-trace --stats of
# tools/c2_trace.py: 400 configs[1]-shaped steps, warm clocks).  Timing-only ablation builds return wrong results; their repairs confound
# step times, the non-gated scan kernel's duration is not affected.     bash tools/ab_regq_kernel.sh "tree ab_x.so" [dim]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LIBS=$1; DIM=${2:-128}
cp $R/zvdb_amd/libzvdb_hip.so /tmp/ab_tree.so
trap 'cp /tmp/ab_tree.so $R/zvdb_amd/libzvdb_hip.so' EXIT
for v in $LIBS; do
  if [ "$v" = tree ]; then cp /tmp/ab_tree.so $R/zvdb_amd/libzvdb_hip.so; else cp "$R/$v" $R/zvdb_amd/libzvdb_hip.so; fi
  rm -rf /tmp/abrq
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abrq -o s -- python3 $R/tools/c2_trace.py 400 $DIM > /tmp/abrq_out.txt 2> /tmp/abrq_err.txt || { tail -5 /tmp/abrq_err.txt; exit 1; }
  python3 - "$v" "$DIM" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/abrq/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_f32_regq" in r["Name"] and int(r["Calls"]) >= 100 and r["Name"].split("k_f32_regq")[1].split("(")[0].endswith("_0"):   # (_1: the gated repair symbol)
        nm = r["Name"].split("k_f32_regq")[1].split("(")[0]
        ms = float(r["AverageNs"]) / 1e6
        print(f"{sys.argv[1]:>20} dim {sys.argv[2]} k_f32_regq{nm}: calls {r['Calls']} avg {ms:.4f} ms min {float(r['MinNs']) / 1e6:.4f} ms = {2.0 * 256 * 1e6 * int(sys.argv[2]) / (ms * 1e-3) / 157.3e12:.3f} of the fp32-MFMA peak", flush=True)
PY
done
