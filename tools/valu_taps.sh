#!/bin/bash
# Where do the tap-list launches' vector instructions go?  SQ_INSTS_VALU / SQ_INSTS_LDS / wave-cycles of the fused
# forward and backward launches with the tap loop ablated (ablation build, DPSX_DBG=16 for the forward kernels;
# 1 | 2 = no mirrored windows for the adjoint): what is left is loader + S1 + epilogue.
#   gpurun -- 'bash tools/valu_taps.sh > gpurun_out/valu_taps.txt 2>&1'
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT/dps_ttc_amd/csrc"
make EXTRA=-DDPSX_ABLATION=1 OBJDIR=../lib/obj_abl OUT=../lib/libdpsx_abl.so > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
export DPSX_LIB=$ROOT/dps_ttc_amd/lib/libdpsx_abl.so
OUT=gpurun_out
for d in ${ABL_SET:-0 16 3}; do
  rm -rf $OUT/valu_$d
  DPSX_DBG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES \
    --kernel-trace --output-format csv -d $OUT/valu_$d -- python3 tools/kbench.py --operator motion_blur --only fwd,bwd,op,score --reps 5 --no-x0 > /dev/null 2> $OUT/valu_$d.err
  echo "== DPSX_DBG=$d"
  python3 - "$OUT/valu_$d" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "taps" not in k or max(len(v) for v in c.values()) < 4:
        continue
    w = sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"])
    print(k)
    print("    per wave: " + "  ".join(f"{n[3:]} {sum(v)/len(v)/w:8.0f}" for n, v in sorted(c.items()) if n != "SQ_WAVES"))
PY
  DPSX_DBG=$d python3 tools/kbench.py --operator motion_blur --only fwd,bwd,op,score --reps 30 --no-x0 2>/dev/null
  rm -rf $OUT/valu_$d
done
