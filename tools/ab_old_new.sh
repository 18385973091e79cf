cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo "== new"; python3 tools/kbench.py --operator gaussian_blur --only fwd,bwd --reps 50 | grep -E "^fwd|^bwd"
  echo "== old"; DPSX_LIB=$GRAFT_REPO_ROOT/dps_ttc_amd/lib/libdpsx_old.so python3 tools/kbench.py --operator gaussian_blur --only fwd,bwd --reps 50 | grep -E "^fwd|^bwd"
done
