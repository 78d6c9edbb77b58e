#!/bin/bash
# Run ON THE GPU BOX: LDS / VALU / wait counters of k_wino4 on the four big shapes (VERDICT r5 "Next 2" evidence).
# Separate --pmc passes with --kernel-trace only (scripts/pmc_any.sh).  Output: gpurun_out/${TAG}_pmc_wino4_lds.md
TAG=${1:-r06}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_pmc_wino4_lds.md
{ echo "# k_wino4: LDS / VALU / wait counters per dispatch (rocprofv3 --kernel-trace --pmc <set>, scripts/pmc_any.sh over scripts/time_wino.py)"; echo; } > $OUT
for shape in "512 64" "256 128" "128 256" "512 32"; do
  set -- $shape
  { echo "## B 16, C $1, H $2"; echo '```'; C=$1 H=$2 bash scripts/pmc_any.sh "k_wino4" python scripts/time_wino.py ${LIBNAME}; echo '```'; echo; } >> $OUT 2>&1
done
tail -50 $OUT
