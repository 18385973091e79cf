#!/bin/bash
# Round evidence that needs no kernel change: PMC traffic of every operator's fused launches, SQ counters of the
# motion-blur and phase-retrieval launches, bench lines at the configs' own N and of the sharded workloads (1 GPU).
#   gpurun --timeout 1200 -- 'bash tools/round_evidence.sh r03'
set -o pipefail
TAG=${1:-r03}; OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
bash tools/pmc_traffic.sh $TAG > $OUT/${TAG}_pmc_traffic.log 2>&1; echo "traffic rc=$?"
tail -12 $OUT/${TAG}_pmc_traffic.log
for spec in "motion_blur fwd" "motion_blur bwd" "phase_retrieval fwd" "phase_retrieval bwd"; do
  set -- $spec
  bash tools/pmc_sq.sh $1 $2 ${TAG}_sq_$1_$2 > /dev/null 2>&1; echo "sq $1 $2 rc=$?"
  rm -rf $OUT/${TAG}_sq_$1_$2_a $OUT/${TAG}_sq_$1_$2_b
done
cat $OUT/${TAG}_sq_*_summary.txt
# BASELINE configs at their own N
python3 bench.py --operator super_resolution --particles 16 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_bench_sr4_n16.json 2> $OUT/${TAG}_bench_misc.err; echo "sr16 rc=$?"
python3 bench.py --operator super_resolution --particles 16 --chains 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_bench_sr4_n16_one_chain.json 2>> $OUT/${TAG}_bench_misc.err
python3 bench.py --operator motion_blur --particles 32 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_bench_motion_n32.json 2>> $OUT/${TAG}_bench_misc.err; echo "motion32 rc=$?"
python3 bench.py --operator motion_blur --particles 32 --workload dps_scores --semantic --steps 200 --warmup 20 > $OUT/${TAG}_bench_motion_n32_semantic_scores.json 2>> $OUT/${TAG}_bench_misc.err; echo "motion32 sem rc=$?"
python3 bench.py --operator phase_retrieval --workload resample --steps 200 --warmup 20 > $OUT/${TAG}_bench_phase_resample.json 2>> $OUT/${TAG}_bench_misc.err; echo "phase resample rc=$?"
python3 bench.py --workload search --steps 200 --warmup 20 > $OUT/${TAG}_bench_search.json 2>> $OUT/${TAG}_bench_misc.err; echo "search rc=$?"
for f in sr4_n16 sr4_n16_one_chain motion_n32 motion_n32_semantic_scores phase_resample search; do
  python3 -c "import json;d=json.load(open('$OUT/${TAG}_bench_$f.json'));r=d['roofline'];print('$f', round(d['value']), 'p-s/s', round(d['ms_per_step']*1e3,1),'us/step', {k:round(v*1e3,1) for k,v in r['all_launches_ms'].items()}, 'step_frac', round(r['step_frac_of_hbm_roofline'],3), 'moved', r.get('step_frac_moved'))"
done
