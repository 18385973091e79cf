set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -5
python3 tools/kbench.py --operator motion_blur --only fwd,bwd,op,adj --reps 30 2>&1 | grep -E "fwd|bwd|upd|op |adj"
