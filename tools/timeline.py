#!/usr/bin/env python3
"""Kernel timeline of the last pipeline pass in a rocprofv3 --kernel-trace csv: tools/timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_hist" in r["Kernel_Name"]]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]["Start_Timestamp"]); prev = t0; busy = 0
for r in rows[s:e]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-30s start=%7.1f dur=%6.1f gap=%5.1f grid=%-7s wg=%s" % (n[:30], (st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3, r["Grid_Size_X"], r["Workgroup_Size_X"]))
    prev = en; busy += en - st
print("busy %.1f us, span %.1f us, %d nodes" % (busy / 1e3, (prev - t0) / 1e3, e - s))
