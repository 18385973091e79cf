#!/bin/bash
# The separable scoring launch (search_ddpm): phase ablation (DPSX_DBG 1 = no horizontal pass, 2 = no vertical pass,
# 8 = no epilogue) and the occupancy probe (DPSX_LDS_PAD: extra dynamic LDS -> fewer resident workgroups per CU).
# Needs the ablation build (tools/abl.sh builds it):  gpurun -- 'bash tools/abl.sh > /dev/null; bash tools/abl_score.sh'
set -e
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export DPSX_LIB=$PWD/dps_ttc_amd/lib/libdpsx_abl.so
for d in 0 1 2 3 8 11; do echo "== DBG=$d"; DPSX_DBG=$d python3 tools/kbench.py --operator gaussian_blur --only score --reps 40 2>&1 | grep "^score"; done
for pad in 0 40000 90000; do echo "== pad $pad"; DPSX_LDS_PAD=$pad python3 tools/kbench.py --operator gaussian_blur --only score --reps 40 2>&1 | grep "^score"; done
