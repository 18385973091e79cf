#!/bin/bash
# the separable adjoint with fold terms (DPSX_SEP_ADJ=fold) against the symmetric-taps form (default), same box
python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "gauss or blur or Gauss or fused_step or golden" > gpurun_out/t_blur.log 2>&1 || { tail -n 40 gpurun_out/t_blur.log; exit 1; }
tail -n 2 gpurun_out/t_blur.log
for mode in fold sym fold sym; do
  echo "== $mode"
  DPSX_SEP_ADJ=$mode python tools/kbench.py --operator gaussian_blur --reps 60 --only bwd,adj --no-x0 2>/dev/null | tail -n 2
done
for mode in fold sym; do
  echo "== sigma 5 $mode"
  DPSX_SEP_ADJ=$mode python tools/kbench.py --operator gaussian_blur --sigma 5.0 --reps 60 --only bwd --no-x0 2>/dev/null | tail -n 1
done
for mode in fold sym; do
  echo "== bench $mode"
  DPSX_SEP_ADJ=$mode python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value']), round(r['ms_per_step']*1e3,1), {k:round(v*1e3,1) for k,v in r['roofline']['all_launches_ms'].items()}, round(r['roofline']['one_chain_ms_per_step']*1e3,1))"
done
