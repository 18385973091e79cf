set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in 0 128 0 128; do
  echo "== DBG $d"
  DPSX_DBG=$d python3 tools/kbench.py --only fwd,bwd --reps 40 2>&1 | grep -E "fwd|bwd"
done
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5
