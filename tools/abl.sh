#!/bin/bash
# Phase ablation of the two separable-blur launches (how DESIGN.md's "loads / passes / epilogue" split was measured).
# Builds an ablation variant of the library next to the product one (the hooks are compiled out of the product
# build), then times the forward / backward halves with phases switched off through the DPSX_DBG mask:
#   1 horizontal pass   2 vertical pass   8 epilogue   16 generic folds   32 in-window folds   128 general loader
# Run through gpurun from the repo root:   gpurun -- 'bash tools/abl.sh > gpurun_out/ablation.txt 2>&1'
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT/dps_ttc_amd/csrc"
make EXTRA=-DDPSX_ABLATION=1 OBJDIR=../lib/obj_abl OUT=../lib/libdpsx_abl.so > /dev/null 2>&1
cd "$ROOT"
export DPSX_LIB=$ROOT/dps_ttc_amd/lib/libdpsx_abl.so
for d in 0 1 2 3 8 11 128; do
  echo "== DPSX_DBG=$d"
  DPSX_DBG=$d python3 tools/kbench.py --only fwd,bwd --reps 40 2>&1 | grep -E "fwd|bwd"
done
unset DPSX_LIB
echo "== product build"
python3 tools/kbench.py --only fwd,bwd,upd --reps 40 2>&1 | grep -E "fwd|bwd|upd"
