#!/bin/bash
# SQ-side counters of one kbench case (where do a kernel's wave-cycles go?):
#   gpurun -- 'bash tools/pmc_sq.sh motion_blur op tag'
set -e -o pipefail
OPER=${1:-motion_blur}; ONLY=${2:-op}; TAG=${3:-sq}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/${TAG}_a -- python3 tools/kbench.py --operator $OPER --only $ONLY --reps 5 --no-x0 > /dev/null 2> $OUT/${TAG}_a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/${TAG}_b -- python3 tools/kbench.py --operator $OPER --only $ONLY --reps 5 --no-x0 > /dev/null 2> $OUT/${TAG}_b.err
python3 - "$OUT/${TAG}_a" "$OUT/${TAG}_b" <<'PY' > $OUT/${TAG}_summary.txt
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if max(len(v) for v in c.values()) < 3:
        continue
    print(k)
    for name, v in sorted(c.items()):
        print(f"    {name:24s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
cat $OUT/${TAG}_summary.txt
