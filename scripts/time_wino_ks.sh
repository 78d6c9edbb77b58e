#!/bin/bash
# Run ON THE GPU BOX: F(4x4) K split (cagc_set_tuning wino4_ks) on the under-filled shapes of per-GPU batch 2 / 4 / 8
cd "$GRAFT_REPO_ROOT"
for shape in "2 512 32" "2 512 64" "4 512 32" "4 512 64" "8 512 32" "2 256 128"; do
  set -- $shape
  for ks in 1 0 2 4 8; do
    echo -n "B $1 C $2 H $3 wino4_ks=$ks: "; B=$1 C=$2 H=$3 CAGC_WINO4_KS=$ks python scripts/time_wino.py 2>&1 | tail -1
  done
done
