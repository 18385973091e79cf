#!/bin/bash
# Occupancy probe of the tap-list kernels: extra dynamic LDS (ablation build) lowers the resident workgroups per CU.
#   gpurun -- 'bash tools/occ_taps.sh > gpurun_out/occ_taps.txt 2>&1'
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT/dps_ttc_amd/csrc"
make EXTRA=-DDPSX_ABLATION=1 OBJDIR=../lib/obj_abl OUT=../lib/libdpsx_abl.so > /dev/null 2>&1
cd "$ROOT"
export DPSX_LIB=$ROOT/dps_ttc_amd/lib/libdpsx_abl.so
for pad in 0 8000 20000 48000 100000; do
  echo "== DPSX_LDS_PAD=$pad"
  DPSX_LDS_PAD=$pad python3 tools/kbench.py --operator motion_blur --only op,score,adj --reps 30 2>&1 | grep -E "^op|^score|^adj"
done
