#!/bin/bash
# particle groups (bench.py --chains k = kernels.ParticleGroups) against one chain, per operator and N
run() { python3 bench.py --operator $1 --particles $2 --chains $3 --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 N=$2 chains=$3', round(r['value']), round(r['ms_per_step']*1e3,1))"; }
for c in 1 2 3 4; do run gaussian_blur 64 $c; done
for c in 1 2 3; do run gaussian_blur 32 $c; done
for c in 1 2 3; do run gaussian_blur 64 $c; done
