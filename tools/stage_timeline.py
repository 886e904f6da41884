#!/usr/bin/env python3
"""The last scoring stage of a rocprofv3 kernel trace, kernel by kernel (start / end in us after the previous kernel's end, queue):
    python tools/stage_timeline.py <p_kernel_trace.csv>"""
import csv
import sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("fseg::", "").replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id"), r.get("Grid_Size"), r.get("LDS_Block_Size")))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_hist") and not r[2].startswith("k_hist_ranges")]
run = rows[idx[-2]:idx[-1]]
STAGE = ("k_solve", "k_dpw", "k_wave", "k_gate", "k_tiny", "k_score", "k_cov", "k_dp")
first = min(i for i, r in enumerate(run) if r[2].startswith(STAGE))
last = max(i for i, r in enumerate(run) if r[2].startswith(STAGE))
# (times count from k_prob_emit's end where the run has one: round 5's waiters, k_wait_word, start before it)
emit = [r[1] for r in run if r[2].startswith("k_prob_emit")]
t0 = max(emit) if emit else run[first - 1][1]
lo = min([first - 1] + [i for i, r in enumerate(run) if r[2].startswith(("k_wait_word", "k_prob_emit"))])
hi = max([last + 2] + [i + 1 for i, r in enumerate(run) if r[2].startswith(("k_signal", "k_segments")) and i <= last + 6])
for s, e, n, q, g, l in run[lo:hi]:
    print("%8.1f %8.1f  %-50s queue %s grid %s lds %s" % ((s - t0) / 1e3, (e - t0) / 1e3, n[:50], q, g, l))
