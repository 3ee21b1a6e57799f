#!/bin/bash
# A/B/... of several builds of libzvdb_hip.so on ONE box for the graph route: tools/ef_probe_r5.py in turn with every library listed
# (round-robin, ROUNDS times); "tree" = zvdb_amd/libzvdb_hip.so as built.   bash tools/ef_ab.sh 2 "tree ab_x.so" [legs]
ROUNDS=$1; LIBS=$2; LEGS=${3:-A,B}
cp zvdb_amd/libzvdb_hip.so /tmp/ab_tree.so
trap 'cp /tmp/ab_tree.so zvdb_amd/libzvdb_hip.so' EXIT
for r in $(seq 1 $ROUNDS); do
  for v in $LIBS; do
    if [ "$v" = tree ]; then cp /tmp/ab_tree.so zvdb_amd/libzvdb_hip.so; else cp "$v" zvdb_amd/libzvdb_hip.so; fi
    timeout -k 10 300 python tools/ef_probe_r5.py "$(basename $v .so)" $LEGS 2>/dev/null || { echo "probe failed for $v"; exit 1; }
  done
done
