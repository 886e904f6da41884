#!/usr/bin/env python3
"""Developer tool: stage times of a batch of DEEP partitions (problems that see 500-1 000 reads: solved by the arena path):
    python tools/deep_probe.py [reads per partition] [partitions]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from freddie_amd import _lib, tables  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
n_part = int(sys.argv[2]) if len(sys.argv) > 2 else 100
params = bench.PARAMS["default"]
tabs = dict(w_main=tables.gaussian_half_kernel(params["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(params["sigma"], 1.0),
            h_table=np.asarray(tables.smooth_threshold(params["threshold_rate"]), np.float64))
w = dict(bench.synth.WORKLOADS["config3"]); w.pop("n_partitions"); w["n_reads"] = n_reads
parts = []
for i in range(n_part):
    g = bench.synth.generate(i, with_seq=False, **w)
    parts.append(bench.pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))
b = bench.Batch(parts)
ctx = _lib.Context(0)
ctx.set_params(**params, **tabs)
ctx.set_profiling(3)
ctx.upload(**b.arrays); ctx.run(); ctx.sync()
acc = {}
for _ in range(10):
    ctx.run(); ctx.sync()
    for k, v in ctx.stage_ms().items():
        acc[k] = acc.get(k, 0.0) + v / 10
print("%d partitions x %d reads: %s" % (n_part, n_reads, ctx.sizes()))
print("  " + " ".join("%s=%.3f" % kv for kv in acc.items() if kv[1] > 0))
ctx.close()
