#!/bin/bash
# kernel-trace of the N = 16 super-resolution step (BASELINE configs[1]) under both forward blockings:
#   gpurun -- 'bash tools/prof_small_n.sh'
set -o pipefail
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in coarse fine; do
  export DPSX_RESIZE_FWD_BLOCKING=$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sn_$mode -- python3 bench.py --operator super_resolution --particles 16 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/sn_$mode.json 2> $OUT/sn_$mode.err
  cp $(ls $OUT/sn_$mode/*/*kernel_stats.csv | head -1) $OUT/sn_${mode}_kernel_stats.csv
  python3 tools/trace_by_grid.py $OUT/sn_$mode > $OUT/sn_${mode}_by_grid.csv
  rm -rf $OUT/sn_$mode
  echo "== $mode"; head -n 8 $OUT/sn_${mode}_by_grid.csv
  python3 -c "import json;r=json.load(open('$OUT/sn_$mode.json'));print(r['value'],r['ms_per_step'],r['roofline']['all_launches_ms'])"
done
