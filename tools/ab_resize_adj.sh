#!/bin/bash
# resize adjoint, same box: fixed-width (ELL) H pass (default) against the CSR form (DPSX_RESIZE_ADJ_CSR=1), and the two
# blockings (DPSX_RESIZE_ADJ_BLOCKING=coarse|fine: 64 / 16 input rows per workgroup)
python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "sr4 or sr8 or resize" 2>&1 | tail -n 1
for n in 64 32 16; do
  echo "== N=$n ELL"; python tools/kbench.py --operator super_resolution --particles $n --reps 60 --only bwd,adj --no-x0 2>/dev/null | tail -n 2
  echo "== N=$n CSR"; DPSX_RESIZE_ADJ_CSR=1 python tools/kbench.py --operator super_resolution --particles $n --reps 60 --only bwd,adj --no-x0 2>/dev/null | tail -n 2
done
for mode in coarse fine; do
  echo "== N=64 ELL $mode"; DPSX_RESIZE_ADJ_BLOCKING=$mode python tools/kbench.py --operator super_resolution --particles 64 --reps 60 --only bwd --no-x0 2>/dev/null | tail -n 1
done
for csr in 0 1; do
  if [ $csr = 1 ]; then export DPSX_RESIZE_ADJ_CSR=1; else unset DPSX_RESIZE_ADJ_CSR; fi
  for n in 64 16; do
    echo "== bench N=$n csr=$csr"
    python bench.py --operator super_resolution --particles $n --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value']), round(r['ms_per_step']*1e3,1), {k:round(v*1e3,1) for k,v in r['roofline']['all_launches_ms'].items()}, round(r['roofline']['one_chain_ms_per_step']*1e3,1))"
  done
done
