#!/usr/bin/env python3
"""Developer tool: where the start-up of a one-GPU run goes (wall time of each step in a fresh process).
    python tools/startup_probe.py"""
import os
import sys
import time

t0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
t1 = time.perf_counter()
from freddie_amd import _lib, tables, pack, synth  # noqa: E402
t2 = time.perf_counter()
L = _lib.load()
t3 = time.perf_counter()
ctx = _lib.Context(0)
t4 = time.perf_counter()
ctx2 = _lib.Context(0)
t5 = time.perf_counter()
params = dict(sigma=5.0, threshold_rate=0.9, variance_factor=3.0, max_problem_size=50, min_read_support_outside=3, ignore_ends=True)
tabs = dict(w_main=tables.gaussian_half_kernel(5.0, 4.0), w_refine=tables.gaussian_half_kernel(5.0, 1.0),
            h_table=np.asarray(tables.smooth_threshold(0.9), np.float64))
ctx.set_params(**params, **tabs)
parts = []
for i in range(20):
    g = synth.generate(i, with_seq=False, n_reads=500, n_exons=40, rp=0.05)
    parts.append(pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))
arrays = pack.concat_batch(parts)
t6 = time.perf_counter()
ctx.upload(**arrays); ctx.run(); ctx.sync()
t7 = time.perf_counter()
ctx.upload(**arrays); ctx.run(); ctx.sync()
t8 = time.perf_counter()
print("numpy import %.3f s | package import %.3f | library load (dlopen) %.3f | first context (HIP start-up, code object, streams, pinned buffers) %.3f | "
      "second context %.3f | synthetic batch %.3f | first upload + run %.3f | second %.3f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7))
ctx.close(); ctx2.close()
