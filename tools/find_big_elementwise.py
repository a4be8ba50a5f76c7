#!/usr/bin/env python
"""development aid: torch's own elementwise / copy / reduce kernels in a rocprofv3 kernel trace of bench.py,
aggregated by (functor, grid size) over the last training step (delimited by adam_kernel launches).

    python tools/find_big_elementwise.py <kernel_trace.csv>
"""
import collections
import csv
import re
import sys


def main():
    rows = []
    with open(sys.argv[1]) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("Grid_Size_X") or r.get("Grid_Size") or "?"))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    seg = rows[ends[-2] + 1: ends[-1] + 1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for st, en, name, grid in seg:
        if "at::native" not in name and "rocclr" not in name:
            continue
        f = re.search(r"(\w+Functor\w*|direct_copy\w*|sum_functor|\w+_kernel_cuda|CatArray\w*|rocclr\w*)", name)
        key = ((f.group(1) if f else name[:50]), grid)
        agg[key][0] += 1
        agg[key][1] += (en - st) / 1e3
    tot = 0.0
    for (fn, grid), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%-34s grid %-10s x%-4d %9.1f us" % (fn[:34], grid, n, us))
        tot += us
    print("total (listed) %.1f us" % tot)


if __name__ == "__main__":
    main()
