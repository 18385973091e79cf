set -e
cd $GRAFT_REPO_ROOT
export DPSX_LIB=$GRAFT_REPO_ROOT/dps_ttc_amd/lib/libdpsx_abl.so
for d in 0 1 2 3 8 11; do echo "== DBG=$d"; DPSX_DBG=$d python3 tools/kbench.py --operator gaussian_blur --only score --reps 40 2>&1 | grep "^score"; done
for pad in 0 40000 90000; do echo "== pad $pad"; DPSX_LDS_PAD=$pad python3 tools/kbench.py --operator gaussian_blur --only score --reps 40 2>&1 | grep "^score"; done
