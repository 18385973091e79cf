set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "phase_spectral" 2>&1 | tail -3
python3 tools/kbench.py --operator phase_retrieval --only fwd,bwd --reps 20 2>&1 | grep -E "fwd|bwd"
