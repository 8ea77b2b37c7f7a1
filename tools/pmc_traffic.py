#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter_collection.csv each) into
profiles/traffic.json: HBM bytes per launch of each kernel, keyed by bench.py's stage names.

    tools/pmc_traffic.py WORKLOAD FETCH.csv WRITE.csv [ROUND_TAG [BATCH]]      (BATCH: frame pairs per launch of the profiled run)

Counter units are KiB.  On gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes, so fetches are doubled;
the correction was checked on hpath_kernel, whose reads are known exactly (two s16 volumes per launch).
"""
import collections
import csv
import json
import os
import re
import sys

STAGE = {"prefilter_kernel": "sgbm_prefilter", "cost_vpath_kernel": "sgbm_cost_vpath", "hpath_ckpt_kernel": "sgbm_hpath_ckpt",
         "hpath_resolve_kernel": "sgbm_hpath_resolve", "lrmedian_kernel": "sgbm_lrcheck_median",
         "lk_row_kernel": "lk"}


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    base, targs = m.group(1), m.group(2) or ""
    if base == "hpath_kernel":
        return "hpath_kernel<true>" if "true" in targs else "hpath_kernel<false>"
    return base


def per_kernel(path, totals=None):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    if totals is not None:   # sum over every dispatch, and how often a once-per-step kernel ran
        totals["sum"] = sum(sum(v) for v in acc.values())
        totals["steps"] = len(acc.get("cost_vpath_kernel", [])) or 1
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    workload, fetch_csv, write_csv = sys.argv[1:4]
    tag = sys.argv[4] if len(sys.argv) > 4 else ""
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else (8 if workload == "4k" else 16)
    tf, tw = {}, {}
    F, W = per_kernel(fetch_csv, tf), per_kernel(write_csv, tw)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_path = os.path.join(root, "profiles", "traffic.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    entry, detail = {}, {}
    for k in sorted(F):
        fetch, write = F[k] * 1024 * 2, W.get(k, 0.0) * 1024
        detail[k] = {"fetch_bytes": fetch, "write_bytes": write}
        if k in STAGE:
            entry[STAGE[k]] = fetch + write
    data[workload] = entry
    data[workload + "_batch"] = batch
    # every kernel of the path, the small ones included: what the whole path moves per launch of one batch
    # (a kernel that runs several times per step, like the pyramid levels, counts with all its launches)
    data[workload + "_all_kernels"] = tf["sum"] * 1024 * 2 / tf["steps"] + tw["sum"] * 1024 / tw["steps"]
    data[workload + "_detail"] = {"per_kernel_per_launch": detail, "source": tag,
                                  "note": "FETCH_SIZE KiB x 1024 x 2 (gfx950 correction), WRITE_SIZE KiB x 1024"}
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
    for k, v in detail.items():
        print("%-24s fetch %10.1f MB  write %10.1f MB" % (k, v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
