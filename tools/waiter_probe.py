#!/usr/bin/env python3
"""Developer tool: which runs of a context lose a device-side waiter (k_wait_word) to its time limit, and how long the waiters sleep.
    python tools/waiter_probe.py [--parts 48] [--runs 6]        (FSEG_NO_SIZED=1 / FSEG_SYNC_TICKS=.. from the environment)
Prints FSEG_TAP_SYNC after every run: [generation, waiters on, emit_gen, side_gen 0, side_gen 1, emit_ctr, time-outs, forked runs].
Under `rocprofv3 --kernel-trace` the trace holds every k_wait_word's duration (tools/waiter_trace.py prints them with their neighbours)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from freddie_amd import _lib, tables  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--parts", type=int, default=48)
ap.add_argument("--runs", type=int, default=6)
ap.add_argument("--workload", default="config4")
args = ap.parse_args()
params = bench.PARAMS["config5" if args.workload == "config5" else "default"]
tabs = dict(w_main=tables.gaussian_half_kernel(params["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(params["sigma"], 1.0),
            h_table=np.asarray(tables.smooth_threshold(params["threshold_rate"]), np.float64))
w = dict(bench.synth.WORKLOADS[args.workload]); w.pop("n_partitions")
parts = []
for i in range(args.parts):
    g = bench.synth.generate(7000 + i, with_seq=False, **w)
    parts.append(bench.pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))
b = bench.Batch(parts)
ctx = _lib.Context(0)
ctx.set_params(**params, **tabs)
ctx.upload(**b.arrays)
for r in range(args.runs):
    t0 = time.perf_counter(); ctx.run(); ctx.sync(); dt = time.perf_counter() - t0
    print("run %d: %.3f ms  sync %s" % (r, dt * 1e3, ctx.tap("sync").tolist()), flush=True)
ctx.close()
