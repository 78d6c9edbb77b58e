#!/bin/bash
# PMC passes over scripts/time_up4.py for the fused-phase kernel (k_conv_up4): per-dispatch counter averages per layer shape.
# Separate --pmc passes, kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE x2 on gfx950; WRITE_SIZE uncalibrated).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export REPS=2
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  rm -rf /tmp/pw; rocprofv3 --kernel-trace --pmc $set -d /tmp/pw -o w -- python scripts/time_up4.py > /dev/null 2>&1
  python - <<'PY'
import sqlite3
cur = sqlite3.connect('/tmp/pw/w_results.db').cursor()
rows = cur.execute("select kernel_name, counter_name, grid_size, count(distinct dispatch_id), sum(value), sum(end-start), min(start) from counters_collection where kernel_name like '%k_conv_up4%' group by kernel_name, counter_name, cast((end-start)/50000 as int) order by counter_name, 7").fetchall()
for kn, n, g, k, v, t, _ in rows: print(f"{kn[20:46]:26s} {n:26s} dispatches {k:4d}  per-dispatch {v/k:12.5g}  avg us {t/k/1e3:8.1f}")
PY
done
