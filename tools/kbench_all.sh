#!/bin/bash
# per-launch micro-bench of every operator and of the search step (step 4 of tools/profile_round.sh)
echo "# tools/kbench.py / tools/kbench_search.py on MI355X, N=64, 256x256, us per launch of the fused step (avg and min over 30)"
echo "# fwd / bwd as the loop launches them: x0_hat not written out (--no-x0); 'with x0_hat store' rows: the API default"
for op in gaussian_blur super_resolution inpainting motion_blur phase_retrieval; do
  echo "== $op"
  python3 tools/kbench.py --operator $op --only fwd,bwd,upd --no-x0 2>/dev/null
done
echo "== gaussian_blur with the x0_hat store"
python3 tools/kbench.py --operator gaussian_blur --only fwd,bwd 2>/dev/null
echo "== gaussian_blur sigma=5.0 (reach 20 px: the 5-tap-group bucket of the separable kernels)"
python3 tools/kbench.py --operator gaussian_blur --sigma 5.0 --only fwd,bwd,upd --no-x0 2>/dev/null
echo "== phase retrieval, the round-2 passes B and C (DPSX_PHASE_V1=1)"
DPSX_PHASE_V1=1 python3 tools/kbench.py --operator phase_retrieval --only fwd,bwd --no-x0 2>/dev/null
echo "== search_ddpm step, replicated form (dpsx_search_step_f32)"
for op in gaussian_blur super_resolution inpainting; do python3 tools/kbench_search.py --operator $op 2>/dev/null; done
echo "== search_ddpm step, ONE state particle (dpsx_search_step_one_f32: what SearchDDPM runs after its first select)"
for op in gaussian_blur super_resolution inpainting; do python3 tools/kbench_search.py --operator $op --one 2>/dev/null; done
