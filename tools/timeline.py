#!/usr/bin/env python3
"""Diagnostic: the timeline of one step from a rocprofv3 kernel trace (python tools/timeline.py <csv> [anchor kernel] [which]).
A step is cut at every dispatch of the anchor kernel (default k_hist); prints start offset and duration of every dispatch of
the chosen step (default: the last but one), so that what really overlaps on the side streams can be seen."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("fseg::", "").replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r["Queue_Id"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["LDS_Block_Size"]), int(r["VGPR_Count"])))
rows.sort()
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_hist"
cuts = [i for i, r in enumerate(rows) if r[2] == anchor]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
a = cuts[which]; b = cuts[which + 1] if which + 1 < 0 or which + 1 < len(cuts) else len(rows)
if which == -1: b = len(rows)
t0 = rows[a][0]
print("%-30s %3s %6s %7s %5s %9s %9s %9s" % ("kernel", "q", "blocks", "lds", "vgpr", "start_us", "dur_us", "end_us"))
for s, e, n, q, g, l, v in rows[a:b]:
    print("%-30s %3d %6d %7d %5d %9.1f %9.1f %9.1f" % (n[:30], q, g, l, v, (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3))
