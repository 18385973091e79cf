#!/bin/bash
# copy the judged summaries of a profile round (tools/profile_round.sh <tag>) from gpurun_out/ (scratch) into profiles/
TAG=${1:-r03}
cd "$(dirname "$0")/.."
for f in gpurun_out/${TAG}_bench_*.json gpurun_out/${TAG}_bench_kernel_stats.csv gpurun_out/${TAG}_bench_kernel_trace_by_grid.csv \
         gpurun_out/${TAG}_operators_kbench.txt gpurun_out/${TAG}_sq_*_summary.txt gpurun_out/${TAG}_traffic_*.json \
         gpurun_out/${TAG}_rccl_one_rank_probe.txt gpurun_out/${TAG}_e2e_unet.txt; do
  [ -s "$f" ] && cp "$f" profiles/
done
[ -s gpurun_out/${TAG}_traffic.json ] && cp gpurun_out/${TAG}_traffic.json profiles/traffic.json
# superseded scratch names of earlier sessions
rm -f profiles/${TAG}_bench_search.json profiles/${TAG}_bench_sr4_n16.json profiles/${TAG}_bench_sr4_n16_one_chain.json \
      profiles/${TAG}_bench_phase_resample.json profiles/${TAG}_bench_motion_n32_semantic_scores.json
ls profiles | grep -c "^${TAG}_"
