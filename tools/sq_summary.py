#!/usr/bin/env python3
"""Per-kernel means of the SQ counters of a rocprofv3 --pmc pass (last dispatches of each kernel).
usage: sq_summary.py <counter_collection.csv> [kernel substring]"""
import csv, sys
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(vals):
    if pat and pat not in k:
        continue
    print(k)
    for cn, v in sorted(vals[k].items()):
        v = v[-3:]
        print("   %-26s %16.0f" % (cn, sum(v) / len(v)))
