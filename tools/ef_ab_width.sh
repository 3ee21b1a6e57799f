#!/bin/bash
# one box: the r5 base build at width 4 against the tree at widths 4 and 8 (tools/ef_probe_r5.py leg A), ROUNDS times round-robin
ROUNDS=${1:-2}; LEGS=${2:-A}
cp zvdb_amd/libzvdb_hip.so /tmp/ab_tree.so
trap 'cp /tmp/ab_tree.so zvdb_amd/libzvdb_hip.so' EXIT
for r in $(seq 1 $ROUNDS); do
  cp ab_efbase.so zvdb_amd/libzvdb_hip.so; EF_PROBE_WIDTH=4 timeout -k 10 300 python tools/ef_probe_r5.py base_w4 $LEGS 2>/dev/null
  cp /tmp/ab_tree.so zvdb_amd/libzvdb_hip.so; EF_PROBE_WIDTH=4 timeout -k 10 300 python tools/ef_probe_r5.py tree_w4 $LEGS 2>/dev/null
  EF_PROBE_WIDTH=8 timeout -k 10 300 python tools/ef_probe_r5.py tree_w8 $LEGS 2>/dev/null
done
