#!/bin/bash
# Only the PMC passes of tools/profile_round.sh: HBM traffic of the three fused launches as the loop issues them (K1 without
# the x0_hat store), FETCH_SIZE / WRITE_SIZE / the L2 fabric read requests in separate rocprofv3 runs, summarised by
# tools/parse_pmc.py.   gpurun -- 'bash tools/pmc_traffic.sh'   -> gpurun_out/r02_traffic_summary.json
set -e -o pipefail
TAG=r02; OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_rdreq
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -- python3 tools/kbench.py --only fwd,bwd,upd --reps 5 --no-x0 > /dev/null 2> $OUT/${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -- python3 tools/kbench.py --only fwd,bwd,upd --reps 5 --no-x0 > /dev/null 2> $OUT/${TAG}_write.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/${TAG}_rdreq -- python3 tools/kbench.py --only fwd,bwd,upd --reps 5 --no-x0 > /dev/null 2> $OUT/${TAG}_rdreq.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats2 -- python3 tools/kbench.py --only fwd,bwd,upd --reps 30 --no-x0 > /dev/null 2> $OUT/${TAG}_stats2.err
python3 tools/parse_pmc.py $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_stats2 $OUT/${TAG}_rdreq > $OUT/${TAG}_traffic_summary.json
cat $OUT/${TAG}_traffic_summary.json
