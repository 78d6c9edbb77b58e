#!/bin/bash
# Build timing-only variants of the Winograd-domain fused-phase kernel: scripts/build_wino4_variants.sh NAME "-DFLAG=.. -DFLAG2=.." [NAME2 "flags" ...]
# -> content-aware-gan-compression_amd/cagc/libcagc_hip_NAME.so (same objects as the product library except conv_up25.o)
set -e
cd "$(dirname "$0")/../content-aware-gan-compression_amd/csrc"
mkdir -p build_alt
OBJS=$(ls build/*.o | grep -v conv_up25.o)
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $FLAGS -c conv_up25.hip -o build_alt/conv_up25_$NAME.o
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build_alt/conv_up25_$NAME.o -o ../cagc/libcagc_hip_$NAME.so
  echo built libcagc_hip_$NAME.so "($FLAGS)"
done
