#!/bin/bash
# Ablations of the tap-list kernels (ablation build; results are wrong, timing only).
#   forward kernels (k_blur_taps), DPSX_DBG bits: 4 = no window reads, 8 = no FMAs, 16 = no tap loop at all,
#   32 = no run-record fetches inside the loop (the first pair of a class is reused), 64 = scalar-load records
#   one-launch adjoint (k_blur_taps_adj): 1 = no mirrored-row passes, 2 = no mirrored-column passes
#   gpurun -- 'bash tools/abl_taps.sh > gpurun_out/abl_taps.txt 2>&1'
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT/dps_ttc_amd/csrc"
make EXTRA=-DDPSX_ABLATION=1 OBJDIR=../lib/obj_abl OUT=../lib/libdpsx_abl.so > /dev/null 2>&1
cd "$ROOT"
export DPSX_LIB=$ROOT/dps_ttc_amd/lib/libdpsx_abl.so
for d in ${ABL_FWD:-0 64 12 44 76 108 16}; do
  echo "== forward kernels DPSX_DBG=$d"
  DPSX_DBG=$d python3 tools/kbench.py --operator motion_blur --only op,score,fwd --reps 40 2>&1 | grep -E "^op|^score|^fwd"
done
for d in ${ABL_ADJ:-0 64 1 2 3}; do
  echo "== adjoint DPSX_DBG=$d"
  DPSX_DBG=$d python3 tools/kbench.py --operator motion_blur --only adj,bwd --reps 40 2>&1 | grep -E "^adj|^bwd"
done
