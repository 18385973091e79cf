#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r03'
# 1. kernel-trace + stats of the SAME command as the headline bench (short run), split by grid size;
# 2. HBM traffic of the three fused launches of EVERY operator from the L2 fabric counters, in separate --pmc passes
#    (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md, HBM section) -> <tag>_traffic.json;
# 3. SQ counters of the motion-blur, phase-retrieval, super-resolution and Gaussian launches (what binds them);
# 4. the per-launch micro-bench of every operator and the search step;
# 5. bench lines: headline (grouped + one chain in one line), every operator, the configs' own N, the sharded workloads on
#    one GPU, a 2-rank self-launched rehearsal (gloo on this box's one GPU).
# Outputs land in gpurun_out/<tag>_*; copy the summaries you want judged into profiles/.
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
B="python3 bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err
cp $(ls $OUT/${TAG}_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
python3 tools/trace_by_grid.py $OUT/${TAG}_stats > $OUT/${TAG}_bench_kernel_trace_by_grid.csv
rm -rf $OUT/${TAG}_stats
echo "[1] stats done"
bash tools/pmc_traffic.sh $TAG > $OUT/${TAG}_pmc_traffic.log 2>&1; echo "[2] traffic rc=$?"
cp $OUT/${TAG}_traffic.json profiles/traffic.json 2>/dev/null     # (on the box only: the bench lines below read it)
for spec in "motion_blur fwd" "motion_blur bwd" "phase_retrieval fwd" "phase_retrieval bwd" "super_resolution fwd" "super_resolution bwd" "gaussian_blur fwd" "gaussian_blur bwd"; do
  set -- $spec
  bash tools/pmc_sq.sh $1 $2 ${TAG}_sq_$1_$2 > /dev/null 2>&1
  rm -rf $OUT/${TAG}_sq_$1_$2_a $OUT/${TAG}_sq_$1_$2_b $OUT/${TAG}_sq_$1_$2_a.err $OUT/${TAG}_sq_$1_$2_b.err
done
echo "[3] sq done"
bash tools/kbench_all.sh > $OUT/${TAG}_operators_kbench.txt
echo "[4] kbench done"
$B --steps 200 --warmup 20 > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
$B --steps 20 --warmup 5 > $OUT/${TAG}_bench_n1_20steps.json 2>> $OUT/${TAG}_bench_n1.err
$B --steps 200 --warmup 20 --x0-store --no-cpu-baseline > $OUT/${TAG}_bench_n1_x0_store.json 2>> $OUT/${TAG}_bench_n1.err
for op in motion_blur super_resolution inpainting phase_retrieval; do
  $B --operator $op --steps 100 --warmup 10 --cpu-steps 2 --cpu-particles 16 > $OUT/${TAG}_bench_${op}.json 2>> $OUT/${TAG}_bench_n1.err
done
$B --operator super_resolution --particles 16 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_bench_config2_sr4_n16.json 2>> $OUT/${TAG}_bench_n1.err
$B --operator motion_blur --particles 32 --workload dps_scores --semantic --steps 200 --warmup 20 > $OUT/${TAG}_bench_config4_shard_motion_n32.json 2>> $OUT/${TAG}_bench_n1.err
$B --operator motion_blur --particles 32 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_bench_motion_n32.json 2>> $OUT/${TAG}_bench_n1.err
$B --operator phase_retrieval --workload resample --steps 200 --warmup 20 > $OUT/${TAG}_bench_config5_shard_phase_resample.json 2>> $OUT/${TAG}_bench_n1.err
$B --workload search --steps 200 --warmup 20 > $OUT/${TAG}_bench_search_single.json 2>> $OUT/${TAG}_bench_n1.err
$B --workload search --search-form replicated --steps 200 --warmup 20 > $OUT/${TAG}_bench_search_replicated.json 2>> $OUT/${TAG}_bench_n1.err
DPSX_BENCH_BACKEND=gloo $B --gpus 2 --steps 50 --warmup 5 --particles 32 > $OUT/${TAG}_bench_2rank_gloo_rehearsal.json 2> $OUT/${TAG}_bench_2rank.err
DPSX_BENCH_BACKEND=gloo $B --gpus 2 --steps 50 --warmup 5 --operator motion_blur --scaling strong --particles 64 --workload dps_scores --semantic > $OUT/${TAG}_bench_2rank_gloo_config4.json 2>> $OUT/${TAG}_bench_2rank.err
DPSX_BENCH_BACKEND=gloo $B --gpus 2 --steps 50 --warmup 5 --operator phase_retrieval --workload resample --particles 32 > $OUT/${TAG}_bench_2rank_gloo_config5.json 2>> $OUT/${TAG}_bench_2rank.err
# 6. the exchanges through the real backend: a process group of ONE rank over "nccl" (= RCCL) -- what this pool's one-GPU
#    boxes can run of it -- per collective (tools/rccl_probe.py) and as whole sharded workloads; the un-gated real-UNet line
$B --workload search --steps 200 --warmup 20 --force-process-group > $OUT/${TAG}_bench_search_single_rccl_x1.json 2>> $OUT/${TAG}_bench_n1.err
$B --operator phase_retrieval --workload resample --steps 200 --warmup 20 --force-process-group > $OUT/${TAG}_bench_config5_shard_rccl_x1.json 2>> $OUT/${TAG}_bench_n1.err
$B --operator phase_retrieval --workload resample --steps 200 --warmup 20 --force-process-group --resample-fetch selected > $OUT/${TAG}_bench_config5_shard_rccl_x1_selected.json 2>> $OUT/${TAG}_bench_n1.err
$B --operator motion_blur --particles 32 --workload dps_scores --semantic --steps 200 --warmup 20 --force-process-group > $OUT/${TAG}_bench_config4_shard_rccl_x1.json 2>> $OUT/${TAG}_bench_n1.err
python3 tools/rccl_probe.py > $OUT/${TAG}_rccl_one_rank_probe.txt 2> /dev/null
python3 tools/e2e_unet.py > $OUT/${TAG}_e2e_unet.txt 2> /dev/null
echo "[5] bench done"
for f in $OUT/${TAG}_bench_*.json; do
  python3 -c "
import json,sys
try:
    d=json.load(open('$f'))
except Exception as e:
    print('$f', 'UNPARSED', e); sys.exit(0)
r=d['roofline']
print('$f'.split('${TAG}_bench_')[1], round(d['value']), 'p-s/s', round(d['ms_per_step']*1e3,1), 'us/step', {k:round(v*1e3,1) for k,v in r['all_launches_ms'].items()}, 'survey-frac', round(r['step_frac_of_hbm_roofline'],3), 'moved-frac', r.get('step_frac_moved') and round(r['step_frac_moved'],3), 'one-chain', r.get('one_chain_ms_per_step') and round(r['one_chain_ms_per_step']*1e3,1))"
done
cat $OUT/${TAG}_operators_kbench.txt
