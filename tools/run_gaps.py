#!/usr/bin/env python3
"""The last whole run of a rocprofv3 kernel trace, kernel by kernel: start (us after the run's first kernel), gap to the
latest end so far, duration, queue:
    python tools/run_gaps.py <p_kernel_trace.csv>"""
import csv
import sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("fseg::", "").replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id")))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_hist") and not r[2].startswith("k_hist_ranges")]
run = rows[idx[-2]:idx[-1] + 1]
pe = run[0][0]
for s, e, n, q in run:
    print("%8.1f gap %6.1f dur %6.1f q%s %s" % ((s - run[0][0]) / 1e3, (s - pe) / 1e3, (e - s) / 1e3, q, n[:44]))
    pe = max(pe, e)
