#!/bin/bash
# single-state search step: finalisation + select + winner copy in one launch (default) against two (DPSX_SEARCH_ONE_UNFUSED=1)
python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "search or score or resample" 2>&1 | tail -n 1
for op in gaussian_blur super_resolution inpainting; do
  for rep in 1 2; do
    echo "== $op fused"; python3 tools/kbench_search.py --operator $op --one --reps 100 2>/dev/null | tail -n 1
    echo "== $op two launches"; DPSX_SEARCH_ONE_UNFUSED=1 python3 tools/kbench_search.py --operator $op --one --reps 100 2>/dev/null | tail -n 1
  done
done
