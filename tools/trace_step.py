#!/usr/bin/env python
"""One training step's kernel timeline from a rocprofv3 --kernel-trace CSV of bench.py (steps are delimited by
adam_kernel launches): per launch its start offset, duration and the idle gap in front of it; runs of the same
kernel are folded.  Development aid for hunting launch overhead.

    python tools/trace_step.py <kernel_trace.csv> [--step -2] [--fold]
"""
import argparse
import collections
import csv
import sys

from trace_gaps import short


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--step", type=int, default=-2)
    ap.add_argument("--fold", action="store_true")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    s = ends[a.step - 1] + 1 if a.step - 1 >= -len(ends) else 0
    seg = rows[s: ends[a.step] + 1]
    t0 = rows[s - 1][1] if s else seg[0][0]
    cur = t0
    out = []
    for st, en, name in seg:
        out.append((short(name), (st - t0) / 1e3, (en - st) / 1e3, max(0.0, (st - cur) / 1e3)))
        cur = max(cur, en)
    print("launches %d   wall %.3f ms" % (len(out), (cur - t0) / 1e6))
    if not a.fold:
        for n, at, d, g in out:
            print("%9.1f us  %-34s %9.1f us  gap %6.1f" % (at, n, d, g))
        return
    cnt = collections.Counter()
    dur = collections.Counter()
    gap = collections.Counter()
    for n, at, d, g in out:
        cnt[n] += 1
        dur[n] += d
        gap[n] += g
    for n, c in cnt.most_common():
        print("%-40s x%4d  %9.1f us  gaps %8.1f us" % (n, c, dur[n], gap[n]))


if __name__ == "__main__":
    main()
