#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv as a compact table: python profiles/show_stats.py <csv> [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in rows[:top]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("fseg::", "").split("(")[0]
    print("%-22s calls=%-4s avg_us=%10.1f total_ms=%9.3f pct=%6.2f" % (
        name[:22], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
