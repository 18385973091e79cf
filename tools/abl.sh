set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "non_square or resize" 2>&1 | tail -12
