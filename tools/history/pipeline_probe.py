#!/usr/bin/env python3
"""Developer tool: where the host-to-host step goes with K contexts taking turns (bench.py's timed region, with clocks):
per call the wall time of upload / run / results on each context's thread, FSEG_TRACE-free.
    python tools/pipeline_probe.py [--contexts 3] [--steps 48]"""
import argparse, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from freddie_amd import _lib, tables
ap = argparse.ArgumentParser()
ap.add_argument("--contexts", type=int, default=3)
ap.add_argument("--steps", type=int, default=48)
ap.add_argument("--workload", default="config4")
ap.add_argument("--torch", action="store_true", help="import torch and initialise its CUDA context first (what bench.py does)")
ap.add_argument("--profiling", type=int, default=0, help="0 none, 1 events around every stage, 2 around the scoring stage only")
args = ap.parse_args()
if args.torch:
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
params = bench.PARAMS["default"]
tabs = dict(w_main=tables.gaussian_half_kernel(params["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(params["sigma"], 1.0),
            h_table=np.asarray(tables.smooth_threshold(params["threshold_rate"]), np.float64))
batches = bench.build_batches(args.workload, 0, 1)
ctxs = [_lib.Context(0) for _ in range(args.contexts)]
for c in ctxs:
    c.set_params(**params, **tabs)
    c.set_profiling(args.profiling)
order = [i % len(batches) for i in range(args.steps)]
bench.one_shot_steps(ctxs, batches, order[:len(batches) * 2])        # warm-up
rec = [[] for _ in ctxs]
def worker(k):
    ctx = ctxs[k]
    for si in range(k, len(order), len(ctxs)):
        b = batches[order[si]]
        t0 = time.perf_counter(); ctx.upload(**b.arrays)
        t1 = time.perf_counter(); ctx.run()
        t2 = time.perf_counter(); ctx.results(packed=True)
        t3 = time.perf_counter()
        rec[k].append((t0, t1, t2, t3))
th = [threading.Thread(target=worker, args=(k,)) for k in range(len(ctxs))]
t_start = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
wall = time.perf_counter() - t_start
r = np.array([x for k in rec for x in k])
d = np.diff(r, axis=1) * 1e3
print("%d contexts: %.3f ms/step; per call (ms) upload %.2f  run %.2f  results %.2f  = %.2f per batch on its thread"
      % (args.contexts, wall / args.steps * 1e3, *np.median(d, axis=0), np.median(d.sum(axis=1))))
for c in ctxs:
    c.close()
