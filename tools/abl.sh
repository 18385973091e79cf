set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "phase or fused or conditioning" 2>&1 | tail -8
python3 tools/kbench.py --operator phase_retrieval --only fwd,bwd,upd --reps 20 2>&1 | grep -E "fwd|bwd|upd|op |adj"
