#!/bin/bash
# HBM traffic of the three fused launches of EVERY operator as the loop issues them (K1 without the x0_hat store where
# the operator allows it), from the L2's fabric counters: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (they do not fit one pass; MI355X_MICROARCH.md, HBM section), summarised by tools/parse_pmc.py with the gfx950
# corrections and merged into gpurun_out/<tag>_traffic.json (copy to profiles/traffic.json: bench.py reads it).
#   gpurun --timeout 1200 -- 'bash tools/pmc_traffic.sh r03'
set -e -o pipefail
TAG=${1:-r03}; OUT=gpurun_out
OPS=${2:-"gaussian_blur motion_blur super_resolution inpainting phase_retrieval"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
NOTE="HBM bytes per launch (N=64, 256x256) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled (gfx950), KiB -> bytes; fused launches as the loop and bench.py issue them: x0_hat not written out (tools/kbench.py --no-x0; inpainting always writes it); phase retrieval fwd = k_pr_rows_fwd + k_pr_cols, bwd = k_finalize_norm + k_pr_rows_inv; source: tools/pmc_traffic.sh -> ${TAG}_traffic_<operator>.json"
rm -f $OUT/${TAG}_traffic.json
for op in $OPS; do
  rm -rf $OUT/${TAG}_fetch_$op $OUT/${TAG}_write_$op
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch_$op -- python3 tools/kbench.py --operator $op --only fwd,bwd,upd --reps 5 --no-x0 > /dev/null 2> $OUT/${TAG}_fetch_$op.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write_$op -- python3 tools/kbench.py --operator $op --only fwd,bwd,upd --reps 5 --no-x0 > /dev/null 2> $OUT/${TAG}_write_$op.err
  python3 tools/parse_pmc.py $OUT/${TAG}_fetch_$op $OUT/${TAG}_write_$op --operator $op --merge $OUT/${TAG}_traffic.json --note "$NOTE" > $OUT/${TAG}_traffic_$op.json
  echo "== $op"; python3 -c "import json;d=json.load(open('$OUT/${TAG}_traffic_$op.json'));print({k:round(v['hbm_bytes']/1e6,1) for k,v in d.items()})"
  rm -rf $OUT/${TAG}_fetch_$op $OUT/${TAG}_write_$op
done
cat $OUT/${TAG}_traffic.json
