set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in 0 1 2 3 8 11 0; do
  echo "== DBG $d"
  DPSX_DBG=$d python3 tools/kbench.py --only fwd,bwd --reps 40 2>&1 | grep -E "fwd|bwd|upd"
done
