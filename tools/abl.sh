set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15
python3 tools/kbench.py --only fwd,bwd,upd --reps 40 2>&1 | grep -E "fwd|bwd|upd"
