set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8
python3 bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
cat gpurun_out/bench_n1.json
