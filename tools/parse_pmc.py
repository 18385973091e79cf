#!/usr/bin/env python3
"""Turn the rocprofv3 counter CSVs of tools/profile_round.sh into per-launch HBM bytes.

    python tools/parse_pmc.py <fetch_dir> <write_dir> [--stats <dir>] [--rdreq <dir>] [--operator NAME --merge profiles/traffic.json]

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts
exactly half of the bytes of a wide (16 B / lane) coalesced read -> doubled here; WRITE_SIZE is exact for
16 B / lane streaming stores.  Prints a JSON object {role: {"kernels", "fetch_bytes", "write_bytes", "hbm_bytes"}}
keyed by the fused launch the kernels belong to (fwd / bwd / upd; phase retrieval's forward half is two kernels, its
backward half includes the small norm finalisation).  Only kernels with >= 4 profiled calls count (tools/kbench.py runs
every fused launch 8 times; the one-off set-up launches of the same tool are left out).
With --rdreq (TCC_EA0_RDREQ_{32B,64B,128B}_sum) it adds "fetch_bytes_by_request_size", the exact fabric read bytes, as
a cross-check of the doubled FETCH_SIZE.  --merge writes {operator: {fwd, bwd, upd}} (bytes per launch at the profiled
N) into the JSON file bench.py reads its `roofline.traffic` from.
"""
import argparse
import collections
import csv
import glob
import json

ROLE = [("k_blur_sep_fwd", "fwd"), ("k_blur_taps<true", "fwd"), ("k_resize_fwd", "fwd"), ("k_mask_step_fwd", "fwd"),
        ("k_pr_rows_fwd", "fwd"), ("k_pr_cols", "fwd"),
        ("k_blur_sep_adj", "bwd"), ("k_blur_taps_adj", "bwd"), ("k_resize_adj", "bwd"), ("k_mask_step_bwd", "bwd"),
        ("k_pr_rows_inv", "bwd"), ("k_finalize_norm", "bwd"),
        ("k_step_update", "upd")]
MIN_CALLS = 4


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
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--stats")
    ap.add_argument("--rdreq")
    ap.add_argument("--operator")
    ap.add_argument("--merge")
    ap.add_argument("--note", default=None)
    a = ap.parse_args()
    fetch, write = counters(a.fetch_dir, "FETCH_SIZE"), counters(a.write_dir, "WRITE_SIZE")
    stats = {}
    if a.stats:
        for f in glob.glob(f"{a.stats}/**/*kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                stats[r["Name"]] = float(r["AverageNs"])
    rd = {}
    if a.rdreq:
        for size, cname in ((32, "TCC_EA0_RDREQ_32B_sum"), (64, "TCC_EA0_RDREQ_64B_sum"), (128, "TCC_EA0_RDREQ_128B_sum")):
            for name, vals in counters(a.rdreq, cname).items():
                rd[name] = rd.get(name, 0.0) + size * sum(vals) / len(vals)
    res = {}
    for name in sorted(set(fetch) | set(write)):
        role = role_of(name)
        calls = max(len(fetch.get(name, [])), len(write.get(name, [])))
        if role is None or calls < MIN_CALLS:
            continue
        fb = 2.0 * 1024.0 * sum(fetch.get(name, [0])) / max(len(fetch.get(name, [0])), 1)
        wb = 1024.0 * sum(write.get(name, [0])) / max(len(write.get(name, [0])), 1)
        e = res.setdefault(role, {"kernels": [], "fetch_bytes": 0.0, "write_bytes": 0.0, "hbm_bytes": 0.0})
        e["kernels"].append({"kernel": name, "calls": calls, "fetch_bytes": fb, "write_bytes": wb,
                             "avg_ns": stats.get(name)})
        e["fetch_bytes"] += fb
        e["write_bytes"] += wb
        e["hbm_bytes"] += fb + wb
        if name in rd:
            e["fetch_bytes_by_request_size"] = e.get("fetch_bytes_by_request_size", 0.0) + rd[name]
    print(json.dumps(res, indent=1))
    if a.merge and a.operator:
        try:
            table = json.load(open(a.merge))
        except Exception:
            table = {}
        table[a.operator] = {role: e["hbm_bytes"] for role, e in res.items()}
        if a.note:
            table["_note"] = a.note
        json.dump(table, open(a.merge, "w"), indent=1)


if __name__ == "__main__":
    main()
