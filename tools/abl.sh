set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
bash tools/profile_round.sh r01 > gpurun_out/prof_r01.log 2>&1
python3 bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
cat gpurun_out/bench_n1.json | cut -c1-1800
for op in super_resolution inpainting motion_blur phase_retrieval; do
  echo "== $op"
  python3 tools/kbench.py --operator $op --only fwd,bwd,upd --reps 30 2>&1 | grep -E "fwd|bwd|upd"
done
python3 tools/kbench_search.py 2>&1 | tail -1
