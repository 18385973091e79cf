#!/usr/bin/env python3
"""Turn the rocprofv3 counter CSVs of tools/profile_round.sh into per-launch HBM bytes.

    python tools/parse_pmc.py <fetch_dir> <write_dir> <stats_dir> [<rdreq_dir>]

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts
exactly half of the bytes of a wide (16 B / lane) coalesced read -> doubled here; WRITE_SIZE is exact for
16 B / lane streaming stores.  Prints a JSON object {launch: {"fetch_bytes", "write_bytes", "hbm_bytes",
"avg_ns"}} keyed by the fused launch it belongs to (fwd / bwd / upd), plus the raw kernel names.
With <rdreq_dir> (TCC_EA0_RDREQ_{32B,64B,128B}_sum) it adds "fetch_bytes_by_request_size", the exact fabric
read bytes, as a cross-check of the doubled FETCH_SIZE.
"""
import collections
import csv
import glob
import json
import sys

ROLE = [("k_blur_sep_fwd", "fwd"), ("k_blur_taps<true, 1", "fwd"), ("k_resize_fwd", "fwd"), ("k_mask_step_fwd", "fwd"),
        ("k_blur_sep_adj", "bwd"), ("k_blur_taps_adj", "bwd"), ("k_resize_adj", "bwd"), ("k_mask_step_bwd", "bwd"),
        ("k_step_update", "upd"), ("k_finalize_norm", "finalize")]


def role_of(name):
    for key, role in ROLE:
        if key in name:
            return role
    return None


def counters(directory, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    fetch_dir, write_dir, stats_dir = sys.argv[1:4]
    fetch, write = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
    stats = {}
    for f in glob.glob(f"{stats_dir}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            stats[r["Name"]] = float(r["AverageNs"])
    rd = {}
    if len(sys.argv) > 4:
        for size, cname in ((32, "TCC_EA0_RDREQ_32B_sum"), (64, "TCC_EA0_RDREQ_64B_sum"), (128, "TCC_EA0_RDREQ_128B_sum")):
            for name, vals in counters(sys.argv[4], cname).items():
                rd[name] = rd.get(name, 0.0) + size * sum(vals) / len(vals)
    res = {}
    for name in sorted(set(fetch) | set(write)):
        role = role_of(name)
        if role is None:
            continue
        fb = 2.0 * 1024.0 * sum(fetch.get(name, [0])) / max(len(fetch.get(name, [0])), 1)
        wb = 1024.0 * sum(write.get(name, [0])) / max(len(write.get(name, [0])), 1)
        res[role] = {"kernel": name, "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                     "avg_ns_in_bench": stats.get(name)}
        if name in rd:
            res[role]["fetch_bytes_by_request_size"] = rd[name]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
