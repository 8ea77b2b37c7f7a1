#!/usr/bin/env python3
"""Summarises one rocprofv3 --pmc pass of SQ counters (counter_collection.csv) per kernel:

    tools/pmc_sq.py counter_collection.csv out.json

Collected with
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU \
              SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 bench.py --steps 3 --warmup 1 --no-cpu --inflight 1

valu_active   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   share of resident-wave cycles spent issuing vector ALU work
wait_any      = SQ_WAIT_ANY / SQ_WAVE_CYCLES           share parked on s_waitcnt / barriers
wait_inst     = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES      share stalled on issue
lds_conflict  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE   extra LDS cycles per LDS cycle
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.match(r"[A-Za-z0-9_]+", name).group(0)


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    out = {}
    for k, c in acc.items():
        n = max(len(calls[k]), 1)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        row = {"launches": n, "valu_insts_per_launch": c.get("SQ_INSTS_VALU", 0.0) / n}
        if wc > 0:
            row.update(valu_active=c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, inst_active=c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                       wait_any=c.get("SQ_WAIT_ANY", 0.0) / wc, wait_inst=c.get("SQ_WAIT_INST_ANY", 0.0) / wc)
        if c.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0:
            row["lds_conflict"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        out[k] = {a: (round(b, 4) if isinstance(b, float) else b) for a, b in row.items()}
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["valu_insts_per_launch"]):
        print("%-24s %s" % (k, v))


if __name__ == "__main__":
    main()
