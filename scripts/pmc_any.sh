#!/bin/bash
# PMC passes over any microbench script:  bash scripts/pmc_any.sh "<kernel name LIKE pattern>" python scripts/time_wgrad.py
# Prints per-dispatch averages of each counter for kernels matching the pattern (separate --pmc passes, kernel-trace only).
PAT="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pw; rocprofv3 --kernel-trace --pmc $set -d /tmp/pw -o w -- "$@" > /dev/null 2>&1
  PAT="$PAT" python - <<'PY'
import os, sqlite3
cur = sqlite3.connect('/tmp/pw/w_results.db').cursor()
rows = cur.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value), sum(end-start) from counters_collection where kernel_name like ? group by kernel_name, counter_name order by kernel_name", ('%' + os.environ['PAT'] + '%',)).fetchall()
for kn, n, k, v, t in rows: print(f"{kn[:60]:60s} {n:28s} dispatches {k:4d}  per-dispatch {v/k:12.5g}  avg us {t/k/1e3:8.1f}")
PY
done
