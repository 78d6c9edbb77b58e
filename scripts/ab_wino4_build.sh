#!/bin/bash
# Run ON THE GPU BOX: a new build of the F(4x4) kernel (cagc/libcagc_hip.so) against the previous one (cagc/libcagc_hip_old.so): bit-identity of
# the two builds and across repetitions, the in-kernel phase trace (cagc/libcagc_hip_trace.so if present), time + clock back to back, then the
# KD step (the arbiter).  bash scripts/ab_wino4_build.sh TAG
TAG=${1:-w4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 400 python scripts/cmp_wino_builds.py libcagc_hip_old.so libcagc_hip.so 2>&1 | tail -3
[ -f content-aware-gan-compression_amd/cagc/libcagc_hip_trace.so ] && C=512 H=64 timeout 120 python scripts/trace_wino4.py 2>/dev/null
timeout 300 python scripts/time_wino_libs.py libcagc_hip_old.so libcagc_hip.so 2>/dev/null
bash scripts/ab_step.sh $TAG 2>/dev/null | tail -4
