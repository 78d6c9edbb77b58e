#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash scripts/profile_step.sh TAG'):  rocprofv3 kernel-trace of the eager KD step + separate
# --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with other trace domains), summarised into gpurun_out/ as
# markdown by scripts/rocpd_stats.py / rocpd_pmc.py.  Copy the summaries you keep into profiles/.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CAGC_OVERLAP_TEACHER=0 CAGC_SIDE_WGRAD=0 CAGC_FORK_TORGB=0   # one stream: per-kernel durations in isolation, as in bench.py's roofline pass
CMD="python bench.py --no-graph --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-full-iteration --no-proxy --no-config3 --sweep 0 $PROFILE_EXTRA"   # PROFILE_EXTRA="--local-batch 2": the per-GPU batch of an 8-GPU run
mkdir -p gpurun_out
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt -- $CMD > gpurun_out/${TAG}_kt.log 2>&1
DB=$(find /tmp/prof_kt -name '*.db' | head -1)
{ echo "command: rocprofv3 --kernel-trace -- $CMD  (CAGC_OVERLAP_TEACHER=0 CAGC_SIDE_WGRAD=0: eager launches on one stream, per-kernel view of the step bench.py times)"; echo;
  python scripts/rocpd_stats.py "$DB" --marker k_gan_kd_loss_tail --last 4 --top 45; } > gpurun_out/${TAG}_kernel_stats.md
if [ -n "$PROFILE_KT_ONLY" ]; then head -60 gpurun_out/${TAG}_kernel_stats.md; exit 0; fi
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc
  rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc -o pmc -- $CMD > gpurun_out/${TAG}_pmc_$C.log 2>&1
  DB=$(find /tmp/prof_pmc -name '*.db' | head -1)
  { echo "command: rocprofv3 --kernel-trace --pmc $C -- $CMD   (counter unit: KB; see bench.py pmc_traffic for the gfx950 correction)"; echo;
    python scripts/rocpd_pmc.py "$DB" --marker k_gan_kd_loss_tail --last 2 --top 30; } > gpurun_out/${TAG}_pmc_$C.md
done
# MFMA-busy pass (north_star: "rocprof HBM GB/s and MFMA-busy"): SQ counters only, own run.  SQ_VALU_MFMA_BUSY_CYCLES counts
# cycles summed over SIMDs..., SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE give the denominators (MI355X_MICROARCH.md, rocprofv3 PMC slots)
rm -rf /tmp/prof_pmc
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/prof_pmc -o pmc -- $CMD > gpurun_out/${TAG}_pmc_MFMA.log 2>&1
DB=$(find /tmp/prof_pmc -name '*.db' | head -1)
{ echo "command: rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- $CMD"; echo;
  python scripts/rocpd_pmc.py "$DB" --marker k_gan_kd_loss_tail --last 2 --top 80 --mfma; } > gpurun_out/${TAG}_pmc_MFMA.md
tail -3 gpurun_out/${TAG}_kt.log | cut -c1-300
head -12 gpurun_out/${TAG}_kernel_stats.md
