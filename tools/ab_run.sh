#!/bin/bash
# A/B/... of several builds of libzvdb_hip.so on ONE box for any probe: runs COMMAND in turn with every library listed (round-robin,
# ROUNDS times), each line of its output prefixed with the build's name; "tree" = zvdb_amd/libzvdb_hip.so as built.
#   bash tools/ab_run.sh 2 "tree ab_x.so" python tools/exact_scan_probe.py 10000000 768
ROUNDS=$1; LIBS=$2; shift 2
cp zvdb_amd/libzvdb_hip.so /tmp/ab_tree.so
trap 'cp /tmp/ab_tree.so zvdb_amd/libzvdb_hip.so' EXIT
for r in $(seq 1 $ROUNDS); do
  for v in $LIBS; do
    if [ "$v" = tree ]; then cp /tmp/ab_tree.so zvdb_amd/libzvdb_hip.so; else cp "$v" zvdb_amd/libzvdb_hip.so; fi
    timeout -k 10 400 "$@" 2>/dev/null | sed "s/^/[$(basename $v .so) r$r] /" || { echo "command failed for $v"; exit 1; }
  done
done
