#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash tools/profile_round.sh r01'
# 1. kernel-trace + stats of the SAME command as the headline bench (short run);
# 2. HBM traffic of the three fused launches from the L2 fabric counters, in separate --pmc passes
#    (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md, HBM section).
# Outputs land in gpurun_out/<tag>_*; copy the summaries you want judged into profiles/.
set -e -o pipefail
TAG=${1:-r01}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -- python3 tools/kbench.py --only fwd,bwd,upd --reps 5 > /dev/null 2> $OUT/${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -- python3 tools/kbench.py --only fwd,bwd,upd --reps 5 > /dev/null 2> $OUT/${TAG}_write.err
# 3. cross-check: the L2's fabric read requests by size (exact bytes = 32 n32 + 64 n64 + 128 n128)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/${TAG}_rdreq -- python3 tools/kbench.py --only fwd,bwd,upd --reps 5 > /dev/null 2> $OUT/${TAG}_rdreq.err
python3 tools/parse_pmc.py $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_stats $OUT/${TAG}_rdreq > $OUT/${TAG}_traffic_summary.txt
cat $OUT/${TAG}_traffic_summary.txt
