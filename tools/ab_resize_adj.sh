#!/bin/bash
# adjoint blockings of the resize kernel (64 / 16 input rows per workgroup), same box
for n in 64 32 16; do for mode in coarse fine; do
  echo "== N=$n $mode"
  DPSX_RESIZE_ADJ_BLOCKING=$mode python tools/kbench.py --operator super_resolution --particles $n --reps 60 --only bwd --no-x0 2>/dev/null | tail -n 1
done; done
for mode in coarse fine; do
  echo "== bench N=64 $mode"
  DPSX_RESIZE_ADJ_BLOCKING=$mode python bench.py --operator super_resolution --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value']), round(r['ms_per_step']*1e3,1), {k:round(v*1e3,1) for k,v in r['roofline']['all_launches_ms'].items()}, round(r['roofline']['one_chain_ms_per_step']*1e3,1))"
done
