#!/bin/bash
# The full -m gpu suite three times back to back on ONE lease: each run is one pytest process (the
# 10M x 768 tests run before the multi-process tests of the same process), the second under PYTHONFAULTHANDLER=1.
# Logs -> gpurun_out/suite_r5/ (copied to profiles/r5/ by hand).
mkdir -p gpurun_out/suite_r5
for i in 1 2 3; do
  if [ $i = 2 ]; then export PYTHONFAULTHANDLER=1; else unset PYTHONFAULTHANDLER; fi
  python -m pytest tests -q -m gpu -x -p no:cacheprovider --durations=8 > gpurun_out/suite_r5/run$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc: $(tail -1 gpurun_out/suite_r5/run$i.log)"
  if [ $rc != 0 ]; then tail -60 gpurun_out/suite_r5/run$i.log; exit $rc; fi
done
