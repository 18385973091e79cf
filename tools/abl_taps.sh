#!/bin/bash
# Ablations of the tap-list kernels (ablation build; results are wrong, timing only).
#   forward kernels (k_blur_taps), DPSX_DBG bit 16 = no tap loop at all (what the load / store phases cost by themselves)
#   one-launch adjoint (k_blur_taps_adj): 1 = no mirrored-row windows, 2 = no mirrored-column strips,
#   4 = the multi-pass fallback instead of the one-scan form
#   gpurun -- 'bash tools/abl_taps.sh > gpurun_out/abl_taps.txt 2>&1'
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT/dps_ttc_amd/csrc"
make EXTRA=-DDPSX_ABLATION=1 OBJDIR=../lib/obj_abl OUT=../lib/libdpsx_abl.so > /dev/null 2>&1
cd "$ROOT"
export DPSX_LIB=$ROOT/dps_ttc_amd/lib/libdpsx_abl.so
for d in ${ABL_FWD:-0 16}; do
  echo "== forward kernels DPSX_DBG=$d"
  DPSX_DBG=$d python3 tools/kbench.py --operator motion_blur --only op,score,fwd --reps 40 2>&1 | grep -E "^op|^score|^fwd"
done
for d in ${ABL_ADJ:-0 1 2 3 4}; do
  echo "== adjoint DPSX_DBG=$d"
  DPSX_DBG=$d python3 tools/kbench.py --operator motion_blur --only adj,bwd --reps 40 2>&1 | grep -E "^adj|^bwd"
done
