#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (last N dispatches of each kernel).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
read, i.e. reports half of such a stream (MI355X_MICROARCH.md, HBM) -- both the raw and the doubled figure are printed.
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [kernel substring]"""
import csv
import sys
from collections import defaultdict


def load(path, counter):
    out = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("fseg::", "").split("(")[0]
            out[name].append(float(r["Counter_Value"]))
    return out


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
pat = sys.argv[3] if len(sys.argv) > 3 else ""
print("%-24s %8s %14s %14s %14s" % ("kernel", "launches", "fetch_KiB(raw)", "fetch_KiB(x2)", "write_KiB"))
for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
    if pat and pat not in k:
        continue
    f = fetch[k][-5:]
    w = write.get(k, [0.0])[-5:]
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    print("%-24s %8d %14.1f %14.1f %14.1f" % (k[:24], len(fetch[k]), fm, 2 * fm, wm))
