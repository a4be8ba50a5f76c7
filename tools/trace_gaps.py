#!/usr/bin/env python
"""Where a training step's wall time goes, from a rocprofv3 --kernel-trace CSV of bench.py:
steps are delimited by adam_kernel launches; per step the union of kernel intervals (busy), the
idle time between kernels, and the idle time attributed to the kernel that FOLLOWS each gap.

    python tools/trace_gaps.py gpurun_out/prof/<host>/<pid>_kernel_trace.csv [--last 5]
"""
import argparse
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(\w+)(?:<[^(]*>)?\(", name)
    base = m.group(1) if m else name[:40]
    if "elementwise" in name or "reduce_kernel" in name or "Functor" in name:
        f = re.search(r"(\w+Functor\w*|direct_copy\w*|sum_functor|FillFunctor)", name)
        base = "torch:" + (f.group(1) if f else base)
    return base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--last", type=int, default=5)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    if len(ends) < a.last + 1:
        sys.exit("only %d adam launches in the trace" % len(ends))
    ends = ends[-(a.last + 1):]
    tot_wall = tot_busy = 0.0
    by_kernel = collections.Counter()
    gap_after = collections.Counter()
    n_launch = 0
    for s in range(a.last):
        seg = rows[ends[s] + 1: ends[s + 1] + 1]
        t0 = rows[ends[s]][1]
        cur_end = t0
        busy = 0
        for st, en, name in seg:
            if st > cur_end:
                gap_after[short(name)] += st - cur_end
                busy += en - st
                cur_end = en
            elif en > cur_end:
                busy += en - cur_end
                cur_end = en
            by_kernel[short(name)] += en - st
            n_launch += 1
        tot_wall += cur_end - t0
        tot_busy += busy
    n = float(a.last)
    print("steps analysed: %d   launches/step: %.0f" % (a.last, n_launch / n))
    print("wall %.3f ms/step   busy %.3f ms/step   idle %.3f ms/step" % (tot_wall / n / 1e6, tot_busy / n / 1e6,
                                                                        (tot_wall - tot_busy) / n / 1e6))
    print("\nkernel time per step (ms):")
    for k, v in by_kernel.most_common(25):
        print("  %-40s %8.3f" % (k, v / n / 1e6))
    print("\nidle time before a kernel, by that kernel (ms/step):")
    for k, v in gap_after.most_common(15):
        print("  %-40s %8.3f" % (k, v / n / 1e6))


if __name__ == "__main__":
    main()
