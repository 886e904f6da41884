#!/usr/bin/env python3
"""What one batch costs when it is uploaded once, run once and downloaded once (the CLI's pattern), as opposed to
replaying a resident batch: distinct config4-shaped batches go through fseg_upload -> fseg_run -> fseg_sync ->
fseg_download on one context; per-batch wall times of each call on stderr, medians at the end.

    python tools/oneshot_probe.py [--batches 6] [--partitions 500] [--reads 500] [--workload config4]
FSEG_TRACE=1 additionally makes the library print its own phase timers."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freddie_amd import _lib, pack, synth, tables  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=6)
ap.add_argument("--partitions", type=int, default=500)
ap.add_argument("--workload", default="config4")
ap.add_argument("--results", action="store_true", help="fetch through fseg_results (pinned, zero-copy) instead of fseg_download")
ap.add_argument("--rounds", type=int, default=2, help="passes over the batches (the second one sees sized arenas)")
args = ap.parse_args()

w = dict(synth.WORKLOADS[args.workload])
w.pop("n_partitions")
params = dict(sigma=5.0, threshold_rate=0.9, variance_factor=3.0, max_problem_size=50, min_read_support_outside=3,
              ignore_ends=True)
if args.workload == "config5":
    params.update(sigma=3.0, threshold_rate=0.8)
tabs = dict(w_main=tables.gaussian_half_kernel(params["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(params["sigma"], 1.0),
            h_table=np.asarray(tables.smooth_threshold(params["threshold_rate"]), np.float64))
batches = []
t0 = time.perf_counter()
for b in range(args.batches):
    parts = []
    for i in range(args.partitions):
        g = synth.generate(b * args.partitions + i, with_seq=False, **w)
        parts.append(pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))
    batches.append((pack.concat_batch(parts), sum(p.n_reads for p in parts)))
print("generated %d batches in %.1f s" % (len(batches), time.perf_counter() - t0), file=sys.stderr)

ctx = _lib.Context(0)
ctx.set_params(**params, **tabs)
rows = []
for rnd in range(args.rounds):
    for bi, (arrs, n_reads) in enumerate(batches):
        t = [time.perf_counter()]
        ctx.upload(**arrs); t.append(time.perf_counter())
        ctx.run(); t.append(time.perf_counter())
        ctx.sync(); t.append(time.perf_counter())
        res = ctx.results() if args.results else ctx.download(); t.append(time.perf_counter())
        d = np.diff(t) * 1e3
        rows.append((rnd, d))
        print("round %d batch %d (%d reads, %.1f MB labels): upload %.2f ms, run %.2f ms, sync %.2f ms, download %.2f ms, total %.2f ms"
              % (rnd, bi, n_reads, res[3].nbytes / 1e6, d[0], d[1], d[2], d[3], d.sum()), file=sys.stderr)
for rnd in range(args.rounds):
    m = np.median(np.array([d for r, d in rows if r == rnd]), axis=0)
    print("round %d medians: upload %.2f  run %.2f  sync %.2f  download %.2f  total %.2f ms  (%.2f M reads/s host-to-host)"
          % (rnd, m[0], m[1], m[2], m[3], m.sum(), batches[0][1] / m.sum() / 1e3))
ctx.close()
