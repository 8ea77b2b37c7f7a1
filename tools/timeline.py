#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace (…_kernel_trace.csv) and says how busy the GPU was over the steady part of the run:
the share of time with at least one kernel running, the mean number of kernels running together, and the gaps.
  tools/timeline.py gpurun_out/prof/default_kernel_trace.csv [first_fraction last_fraction]
"""
import csv
import sys
from collections import defaultdict


def kname(n):
    if n.startswith("void "):
        n = n[5:]
    n = n.replace("(anonymous namespace)::", "")
    return n.split("(")[0].split("<")[0]


def main():
    path = sys.argv[1]
    f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kname(r["Kernel_Name"])))
    rows.sort()
    t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
    a, b = t_lo + (t_hi - t_lo) * f0, t_lo + (t_hi - t_lo) * f1
    sel = [(max(s, a), min(e, b), n) for s, e, n in rows if e > a and s < b]
    ev = []
    for s, e, _ in sel:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = 0.0; area = 0.0; depth = 0; last = a; gaps = []
    hist = defaultdict(float)
    for t, d in ev:
        if depth > 0:
            busy += t - last
        elif t - last > 0:
            gaps.append(t - last)
        hist[depth] += t - last
        area += depth * (t - last)
        depth += d; last = t
    span = b - a
    print("window %.1f ms, %d kernels" % (span / 1e6, len(sel)))
    print("busy %.3f  mean kernels in flight while busy %.2f" % (busy / span, area / max(busy, 1)))
    print("time share by number of kernels running:", {k: round(v / span, 3) for k, v in sorted(hist.items())})
    gaps.sort(reverse=True)
    print("idle gaps: %d, total %.2f ms, largest (us): %s" % (len(gaps), sum(gaps) / 1e6, [round(g / 1e3, 1) for g in gaps[:8]]))
    per = defaultdict(lambda: [0, 0.0])
    for s, e, n in sel:
        per[n][0] += 1; per[n][1] += e - s
    for n, (cnt, tot) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  %-40s %5d launches  %8.3f ms total  %7.3f ms avg  %.3f of window" % (n[:40], cnt, tot / 1e6, tot / 1e6 / cnt, tot / span))


if __name__ == "__main__":
    main()
