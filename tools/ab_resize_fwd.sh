set -e
python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "sr4 or sr8 or resize" > gpurun_out/t_sr.log 2>&1 || { tail -n 40 gpurun_out/t_sr.log; exit 1; }
tail -n 2 gpurun_out/t_sr.log
for n in 16 32 40 48; do
  for mode in coarse fine; do
    echo "== N=$n $mode"
    DPSX_RESIZE_FWD_BLOCKING=$mode python tools/kbench.py --operator super_resolution --particles $n --reps 50 --only fwd,op,score --no-x0 2>&1 | grep -v "^$" | tail -n 4
  done
done
