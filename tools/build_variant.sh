#!/bin/bash
# A/B builds of one translation unit of libzvdb_hip.so (how the kernel variants of profiles/r3/bf16_stream_ab.txt were made):
#   bash tools/build_variant.sh NAME "-DFLAGS" [unit]   ->  ./ab_NAME.so   (unit: bf_bf16_stream by default; the other objects are
#   reused from zvdb_amd/csrc/build/, so run `make -C zvdb_amd/csrc` first).  The .so files are git-ignored and travel with gpurun:
#   gpurun -- 'bash tools/ab_bench.sh 3 "tree ab_NAME.so" --mode bf16 --no-cpu-baseline --no-paths'
set -e
cd "$(dirname "$0")/../zvdb_amd/csrc"
UNIT=${3:-bf_bf16_stream}
EXTRA=""
case "$UNIT" in exact_kernels|ef_search|rerank|graph_build|exact_scan) EXTRA="-ffp-contract=off";; esac
mkdir -p /tmp/var_$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include $EXTRA $2 -c $UNIT.hip -o /tmp/var_$1/$UNIT.o > /tmp/var_$1/log 2>&1
OBJS=""
for o in api exact_kernels ef_search rerank data_kernels bruteforce bf_fp4_stream bf_i8_stream bf_bf16_stream bf_f32_regq graph_build exact_scan comm; do
  if [ "$o" = "$UNIT" ]; then OBJS="$OBJS /tmp/var_$1/$UNIT.o"; else OBJS="$OBJS build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab_$1.so $OBJS -ldl
echo built ab_$1.so
