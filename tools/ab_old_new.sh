#!/bin/bash
# Same-box A/B of two builds of the library (box-to-box variance on the pool is +-10 %, larger than most kernel changes).
# Build the reference build first, HERE (not on the GPU box), from any commit:
#   rm -rf /tmp/old_src && mkdir -p /tmp/old_src && git archive <commit> dps_ttc_amd/csrc include | tar -x -C /tmp/old_src
#   make -C /tmp/old_src/dps_ttc_amd/csrc OBJDIR=/tmp/old_src/obj OUT=$PWD/dps_ttc_amd/lib/libdpsx_old.so
# then:  gpurun -- 'bash tools/ab_old_new.sh [operator] [cases]'
set -e
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OPER=${1:-gaussian_blur}; ONLY=${2:-fwd,bwd}
for rep in 1 2 3; do
  echo "== new"; python3 tools/kbench.py --operator $OPER --only $ONLY --reps 50 --no-x0 | grep -E "^(fwd|bwd|upd|op|adj|score)"
  echo "== old"; DPSX_LIB=$PWD/dps_ttc_amd/lib/libdpsx_old.so python3 tools/kbench.py --operator $OPER --only $ONLY --reps 50 --no-x0 | grep -E "^(fwd|bwd|upd|op|adj|score)"
done
