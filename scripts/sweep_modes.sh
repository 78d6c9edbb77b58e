cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4 5 6 7 8; do SEGS=10 python scripts/sweep_modes.py 4 2>/dev/null | tail -2 | cut -c1-260; done
