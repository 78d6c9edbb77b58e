#!/bin/bash
# PMC passes over the Winograd microbench (scripts/time_wino.py); prints per-kernel sums for k_wino
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_REQ" "TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pw; rocprofv3 --kernel-trace --pmc $set -d /tmp/pw -o w -- python scripts/time_wino.py > /dev/null 2>&1
  python - <<'PY'
import sqlite3
cur = sqlite3.connect('/tmp/pw/w_results.db').cursor()
rows = cur.execute("select counter_name, count(distinct dispatch_id), sum(value) from counters_collection where kernel_name like '%k_wino<%' group by counter_name").fetchall()
for n, k, v in rows: print(f"{n:32s} dispatches {k:3d}  per-dispatch {v/k:.4g}")
PY
done
