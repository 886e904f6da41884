#!/usr/bin/env python3
"""Developer tool: per-stage device time of one batch, first run (sized, plain launches) and replays.
    python tools/replay_probe.py [--workload config4] [--batch 0] [--reps 20]
FSEG_NO_GRAPH=1 gives per-stage events on the replays too; FSEG_LIB=<other build> compares builds."""
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
ap.add_argument("--workload", default="config4")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--profiling", type=int, default=1, help="0: no stage events (the product's single graph), 1: scoring stage bracketed, 3: every stage")
args = ap.parse_args()
params = bench.PARAMS["config5" if args.workload == "config5" else "default"]
tabs = dict(w_main=tables.gaussian_half_kernel(params["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(params["sigma"], 1.0),
            h_table=np.asarray(tables.smooth_threshold(params["threshold_rate"]), np.float64))
per, _ = bench.plan_batches(args.workload, 1)
w = dict(bench.synth.WORKLOADS[args.workload]); w.pop("n_partitions")
parts = []
for i in range(per):
    g = bench.synth.generate(i, with_seq=False, **w)
    parts.append(bench.pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))
b = bench.Batch(parts)
ctx = _lib.Context(0)
ctx.set_params(**params, **tabs)
ctx.set_profiling(args.profiling)
for rnd in range(2):
    ctx.upload(**b.arrays)
    t0 = time.perf_counter(); ctx.run(); ctx.sync(); t1 = time.perf_counter()
    ms = ctx.stage_ms()
    print("first run %d: %.3f ms wall; stages: %s" % (rnd, (t1 - t0) * 1e3, " ".join("%s=%.3f" % kv for kv in ms.items() if kv[1] > 0)))
alg = ctx.scoring_algorithmic_bytes()
for _ in range(2):                      # (the first replay of a one-stream run captures and instantiates its hipGraph: 3-4 ms, once)
    ctx.run(); ctx.sync()
acc = {}
t0 = time.perf_counter()
t_run = 0.0
for _ in range(args.reps):
    h0 = time.perf_counter(); ctx.run(); t_run += time.perf_counter() - h0
    ctx.sync()
    for k, v in ctx.stage_ms().items():
        acc[k] = acc.get(k, 0.0) + v
dt = (time.perf_counter() - t0) / args.reps * 1e3
sc = acc.get("interval_scoring", 0.0) / args.reps or float("nan")
if args.profiling:
    print("replay: %.3f ms/step, %.1f M reads/s; scoring %.3f ms = %.0f GB/s credited (frac %.3f); sizes %s" % (
        dt, b.n_reads / dt / 1e3, sc, alg / sc / 1e6, alg / sc / 1e6 / 8000.0, ctx.sizes()))
else:
    print("replay: %.3f ms/step, %.1f M reads/s (no stage events)" % (dt, b.n_reads / dt / 1e3))
print("  " + " ".join("%s=%.3f" % (k, v / args.reps) for k, v in acc.items() if v > 0))
print("  host time of fseg_run (enqueue only): %.3f ms" % (t_run / args.reps * 1e3))
ctx.close()
