#!/usr/bin/env python3
"""What the scoring stage takes in a rocprofv3 kernel trace of a run with the DEFAULT plan (streams, gate):
    python tools/stage_span.py <p_kernel_trace.csv>
Per run (a run starts with k_hist) two spans, medians over the runs:
  kernels   first start -> last end of the stage's kernels (k_solve, k_dpw, k_wave / k_tiny, k_gate, k_cover, k_score*)
  bracket   end of the last kernel before the stage -> start of the first kernel after it: what HIP events recorded on the
            stream before and after the stage's launches see (bench.py's roofline.launch_ms), fork / join latencies included
"""
import csv
import sys

import numpy as np

STAGE = ("k_solve", "k_dpw", "k_wave", "k_tiny", "k_gate", "k_cover", "k_score")
SYNC = ("k_wait_word", "k_signal")
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("fseg::", "").replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
runs, cur = [], []
for s, e, n in rows:
    if n.startswith("k_hist") and not n.startswith("k_hist_ranges") and cur:
        runs.append(cur); cur = []
    cur.append((s, e, n))
if cur:
    runs.append(cur)
ker, brk, names = [], [], set()
for run in runs:
    st = [(s, e, n) for s, e, n in run if n.startswith(STAGE)]
    if not st:
        continue
    first, last = min(s for s, _, _ in st), max(e for _, e, _ in st)
    # (round 5: the side streams' waiters, k_wait_word, start long before the stage and k_signal / the join's waiter are not
    # stage work: the bracket is named -- k_prob_emit's end to k_segments' start -- where the run has both)
    emit = [e for s, e, n in run if n.startswith("k_prob_emit")]
    segs = [s for s, e, n in run if n.startswith("k_segments")]
    if emit and segs:
        before, after = [max(emit)], [min(x for x in segs if x >= max(emit))] if any(x >= max(emit) for x in segs) else []
    else:
        before = [e for s, e, n in run if e <= first and not n.startswith(STAGE + SYNC)]
        after = [s for s, e, n in run if s >= last and not n.startswith(STAGE + SYNC)]
    ker.append((last - first) / 1e3)
    if before and after:
        brk.append((min(after) - max(before)) / 1e3)
    names.update(n for _, _, n in st)
print("runs with a scoring stage: %d; its kernels: %s" % (len(ker), ", ".join(sorted(names))))
if ker:
    print("kernels  (first start -> last end):                 median %.1f us  min %.1f  max %.1f" % (np.median(ker), min(ker), max(ker)))
if brk:
    print("bracket  (previous kernel's end -> next kernel's start): median %.1f us  min %.1f  max %.1f" % (np.median(brk), min(brk), max(brk)))
