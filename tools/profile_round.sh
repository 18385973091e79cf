#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02'
# 1. kernel-trace + stats of the SAME command as the headline bench (short run);
# 2. HBM traffic of the three fused launches from the L2 fabric counters, in separate --pmc passes
#    (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md, HBM section);
# 3. cross-check: the L2's fabric read requests by size (exact bytes = 32 n32 + 64 n64 + 128 n128);
# 4. the per-launch micro-bench of every operator, the search step, a wide Gaussian (sigma = 5) and a 2-rank
#    self-launched bench rehearsal (gloo on this box's one GPU).
# Outputs land in gpurun_out/<tag>_*; copy the summaries you want judged into profiles/.
set -e -o pipefail
TAG=${1:-r02}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -- python3 tools/kbench.py --only fwd,bwd,upd --reps 5 --no-x0 > /dev/null 2> $OUT/${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -- python3 tools/kbench.py --only fwd,bwd,upd --reps 5 --no-x0 > /dev/null 2> $OUT/${TAG}_write.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/${TAG}_rdreq -- python3 tools/kbench.py --only fwd,bwd,upd --reps 5 --no-x0 > /dev/null 2> $OUT/${TAG}_rdreq.err
python3 tools/parse_pmc.py $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_stats $OUT/${TAG}_rdreq > $OUT/${TAG}_traffic_summary.json
cp $(ls $OUT/${TAG}_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
# the same trace split by grid size: one chain of N particles (what roofline.avg_launch_ms is measured on) vs the timed
# loop's groups of N / chains particles, which run beside each other on their own streams
python3 tools/trace_by_grid.py $OUT/${TAG}_stats > $OUT/${TAG}_bench_kernel_trace_by_grid.csv
{
  echo "# tools/kbench.py / tools/kbench_search.py on MI355X, N=64, 256x256, us per launch of the fused step (avg and min over 30)"
  for op in gaussian_blur super_resolution inpainting motion_blur phase_retrieval; do
    echo "== $op"
    python3 tools/kbench.py --operator $op --only fwd,bwd,upd 2>/dev/null
  done
  echo "== gaussian_blur sigma=5.0 (reach 20 px: the 5-tap-group bucket of the separable kernels)"
  python3 tools/kbench.py --operator gaussian_blur --sigma 5.0 --only fwd,bwd,upd 2>/dev/null
  echo "== search_ddpm step"
  for op in gaussian_blur super_resolution inpainting; do python3 tools/kbench_search.py --operator $op 2>/dev/null; done
  echo "== search_ddpm step, the loop's state held as ONE particle (dpsx_search_step_one_f32: what SearchDDPM runs after its first select)"
  for op in gaussian_blur super_resolution inpainting; do python3 tools/kbench_search.py --operator $op --one 2>/dev/null; done
} > $OUT/${TAG}_operators_kbench.txt
python3 bench.py --steps 200 --warmup 20 > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_n1_20steps.json 2>> $OUT/${TAG}_bench_n1.err
python3 bench.py --steps 200 --warmup 20 --chains 1 --no-cpu-baseline > $OUT/${TAG}_bench_n1_one_chain.json 2>> $OUT/${TAG}_bench_n1.err
python3 bench.py --steps 200 --warmup 20 --x0-store --no-cpu-baseline > $OUT/${TAG}_bench_n1_x0_store.json 2>> $OUT/${TAG}_bench_n1.err
python3 bench.py --steps 200 --warmup 20 --x0-store --chains 1 --no-cpu-baseline > $OUT/${TAG}_bench_n1_one_chain_x0_store.json 2>> $OUT/${TAG}_bench_n1.err
for op in motion_blur super_resolution inpainting phase_retrieval; do
  python3 bench.py --operator $op --steps 100 --warmup 10 --cpu-steps 2 --cpu-particles 16 > $OUT/${TAG}_bench_${op}.json 2>> $OUT/${TAG}_bench_n1.err
done
DPSX_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 50 --warmup 5 --particles 32 > $OUT/${TAG}_bench_2rank_gloo_rehearsal.json 2> $OUT/${TAG}_bench_2rank.err
cat $OUT/${TAG}_operators_kbench.txt
tail -c 1500 $OUT/${TAG}_bench_n1.json
