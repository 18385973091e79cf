set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "resize or sr4 or sr8 or fused or golden" 2>&1 | tail -5
python3 tools/kbench.py --operator super_resolution --only fwd,bwd,op,adj --reps 30 2>&1 | grep -E "fwd|bwd|upd|op |adj"
DPSX_RESIZE_NO_ROWS=1 python3 tools/kbench.py --operator super_resolution --only fwd,op --reps 30 2>&1 | grep -E "fwd|bwd|upd|op |adj"
