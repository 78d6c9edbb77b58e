#!/bin/bash
# Build a timing-only variant of ONE kernel file: scripts/build_variant.sh FILE(no .hip) NAME "-DFLAG=.."
# -> content-aware-gan-compression_amd/cagc/libcagc_hip_NAME.so (same objects as the product library except FILE.o)
set -e
cd "$(dirname "$0")/../content-aware-gan-compression_amd/csrc"
mkdir -p build_alt
F=$1; NAME=$2; FLAGS=$3
OBJS=$(ls build/*.o | grep -v "/$F.o")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $FLAGS -c $F.hip -o build_alt/${F}_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build_alt/${F}_$NAME.o -o ../cagc/libcagc_hip_$NAME.so
echo built libcagc_hip_$NAME.so "($FLAGS)"
