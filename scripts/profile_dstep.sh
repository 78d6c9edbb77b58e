#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace of the discriminator step and of the whole training iteration (row f1, train.py:241-278,371-398)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for what in d r1 g; do
  rm -rf /tmp/prof_d
  rocprofv3 --kernel-trace -d /tmp/prof_d -o kt -- python scripts/dstep.py $what > gpurun_out/${TAG}_dstep_$what.log 2>&1
  DB=$(find /tmp/prof_d -name '*.db' | head -1)
  { echo "command: rocprofv3 --kernel-trace -- python scripts/dstep.py $what   (d = discriminator step of the training iteration, bs 16, eager: D forward on fake + real, D backward incl. every weight gradient + Adam; r1 = the R1 regulariser's double backward; g = the path-length regulariser; whole process: 2 warm-up + 4 steps)"; echo;
    python scripts/rocpd_stats.py "$DB" --top 40; } > gpurun_out/${TAG}_dstep_${what}_kernel_stats.md
  head -14 gpurun_out/${TAG}_dstep_${what}_kernel_stats.md | cut -c1-160
done
