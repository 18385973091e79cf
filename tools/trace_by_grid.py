#!/usr/bin/env python3
"""Per-(kernel, grid size) launch statistics from a rocprofv3 --kernel-trace CSV.
`rocprofv3 --stats` averages all launches of a kernel NAME; bench.py launches the same kernels in two shapes -- one chain
of all N particles, back to back (the launches `roofline.avg_launch_ms` is measured on) and, in the timed loop, one group
of N / chains particles per HIP stream running beside the other groups' launches -- so the table is split by grid size.
    tools/trace_by_grid.py <dir with *_kernel_trace.csv> [name-filter]  > profiles/rNN_bench_kernel_trace_by_grid.csv"""
import collections
import csv
import glob
import re
import sys


def main():
    root = sys.argv[1]
    flt = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"dpsx::")
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not flt.search(name):
                continue
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            acc[(re.sub(r"\(.*", "", name), grid // max(wg, 1))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Workgroups", "Calls", "AverageNs", "MinNs", "MaxNs"])
    for (name, blocks), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, blocks, len(v), round(sum(v) / len(v), 1), min(v), max(v)])


if __name__ == "__main__":
    main()
