#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace of the HIP-graph-replayed KD step at a per-GPU batch (default 2), summarised as a timeline
# (kernels in flight, gaps) by scripts/rocpd_timeline.py:  bash scripts/profile_graph_timeline.sh TAG [BATCH]
TAG=${1:-r06}; LB=${2:-2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
CMD="python bench.py --graph --local-batch $LB --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-full-iteration --no-proxy --no-config3 --sweep 0"
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace -d /tmp/prof_tl -o kt -- $CMD > gpurun_out/${TAG}_tl.log 2>&1
DB=$(find /tmp/prof_tl -name '*.db' | head -1)
{ echo "command: rocprofv3 --kernel-trace -- $CMD (HIP-graph replay, teacher overlap on)"; echo;
  python scripts/rocpd_timeline.py "$DB" --marker k_gan_kd_loss_tail --last 4; } > gpurun_out/${TAG}_graph_bs${LB}_timeline.md
head -12 gpurun_out/${TAG}_graph_bs${LB}_timeline.md | cut -c1-220
