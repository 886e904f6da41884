#!/usr/bin/env python3
"""Per-kernel medians of a rocprofv3 *_kernel_trace.csv: python profiles/trace_medians.py <csv>
The --stats average of a bench run is skewed by the first (arena-sizing) runs of a batch, where some kernels see
empty or oversized lists; the median over the calls is the steady-state duration."""
import collections
import csv
import sys

d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("fseg::", "").replace("void ", "").split("(")[0]
    d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-28s %5s %10s %10s %10s" % ("kernel", "calls", "median_us", "min_us", "max_us"))
for k, v in sorted(d.items(), key=lambda kv: -sorted(kv[1])[len(kv[1]) // 2] * len(kv[1])):
    vs = sorted(v)
    print("%-28s %5d %10.1f %10.1f %10.1f" % (k[:28], len(v), vs[len(vs) // 2], vs[0], vs[-1]))
