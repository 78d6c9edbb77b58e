#!/bin/bash
# Run ON THE GPU BOX: F(4x4) kernel parity (scripts/check_wino4.py, tests/test_wino4_gpu.py) and a same-box A/B of two builds
# (cagc/libcagc_hip_old.so vs cagc/libcagc_hip.so) on the four discriminator / teacher shapes.  Output under gpurun_out/.
TAG=${1:-w4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python scripts/check_wino4.py > gpurun_out/${TAG}_check.log 2>&1; tail -12 gpurun_out/${TAG}_check.log
for shape in "512 64" "256 128" "128 256" "512 32"; do
  set -- $shape
  for lib in libcagc_hip_old.so libcagc_hip.so; do
    [ -f content-aware-gan-compression_amd/cagc/$lib ] && C=$1 H=$2 python scripts/time_wino.py $lib
  done
done 2>&1 | tee gpurun_out/${TAG}_time.log
timeout 900 python -m pytest tests/test_wino4_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/${TAG}_pytest.log
