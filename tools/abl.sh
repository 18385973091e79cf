set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  python3 tools/kbench.py --only fwd,bwd,upd --reps 40 2>&1 | grep -E "fwd|bwd|upd"
done
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5
python3 bench.py --no-cpu-baseline
