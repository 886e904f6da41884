#!/usr/bin/env python3
"""Every k_wait_word of a rocprofv3 kernel trace: duration, queue, and for the longest ones the kernels that ran while it slept.
    python tools/waiter_trace.py <p_kernel_trace.csv> [n_longest]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("fseg::", "").replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id")))
rows.sort()
w = [r for r in rows if r[2].startswith("k_wait_word")]
d = sorted((e - s) / 1e3 for s, e, _, _ in w)
if not d:
    print("no k_wait_word in the trace")
    sys.exit(0)
print("k_wait_word: %d calls, median %.1f us, p90 %.1f, max %.1f us, total %.3f ms" % (len(d), d[len(d) // 2], d[int(len(d) * 0.9)], d[-1], sum(d) / 1e3))
for s, e, n, q in sorted(w, key=lambda r: r[0] - r[1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 3]:
    print("-- waiter on queue %s slept %.1f us; kernels that started or ended while it did:" % (q, (e - s) / 1e3))
    for s2, e2, n2, q2 in rows:
        if e2 >= s - 20000 and s2 <= e + 5000 and (s2, e2, n2, q2) != (s, e, n, q):
            print("   %9.1f %9.1f  %-46s queue %s" % ((s2 - s) / 1e3, (e2 - s) / 1e3, n2[:46], q2))
