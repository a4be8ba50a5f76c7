#!/usr/bin/env python
"""Print a window of consecutive kernels from a rocprofv3 --kernel-trace CSV: start offset, duration and the
idle gap before each one (microseconds).  For launch-bound loops such as the greedy decode.

    python tools/trace_window.py <kernel_trace.csv> [--at 0.5] [--n 40]
"""
import argparse
import csv
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--at", type=float, default=0.5, help="position in the trace, 0..1")
    ap.add_argument("--n", type=int, default=40)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    i0 = int(len(rows) * a.at)
    t0 = rows[i0][0]
    prev_end = rows[i0 - 1][1] if i0 else t0
    for st, en, name in rows[i0:i0 + a.n]:
        m = re.search(r"(\w+)(?:<[^(]*>)?\(", name)
        print("%10.1f  dur %7.1f  gap %7.1f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3,
                                                   (m.group(1) if m else name)[:40]))
        prev_end = en


if __name__ == "__main__":
    main()
