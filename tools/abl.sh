set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "persistent or fused_step" 2>&1 | tail -5
for d in 0 256 0 256; do
  echo "== DBG $d"
  DPSX_DBG=$d python3 tools/kbench.py --only bwd --reps 40 2>&1 | grep -E "fwd|bwd|upd"
done
