#!/usr/bin/env python3
"""Developer probe: what does context B's work cost while context A copies results out / uploads?
    python tools/overlap_probe.py"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from freddie_amd import _lib, tables
params = bench.PARAMS["default"]
tabs = dict(w_main=tables.gaussian_half_kernel(5.0, 4.0), w_refine=tables.gaussian_half_kernel(5.0, 1.0), h_table=np.asarray(tables.smooth_threshold(0.9)))
per, _ = bench.plan_batches("config4", 1)
w = dict(bench.synth.WORKLOADS["config4"]); w.pop("n_partitions")
def batch(base):
    parts = []
    for i in range(per):
        g = bench.synth.generate(base + i, with_seq=False, **w)
        parts.append(bench.pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))
    return bench.Batch(parts)
b0, b1 = batch(0), batch(per)
A, B = _lib.Context(0), _lib.Context(0)
for c, b in ((A, b0), (B, b1)):
    c.set_params(**params, **tabs); c.upload(**b.arrays); c.run(); c.sync(); c.run(); c.sync()
stop = False
def a_results():
    while not stop:
        A.run(); A.results()
def a_copy_only():
    while not stop:
        A.results()
def a_run_only():
    while not stop:
        A.run(); A.sync()
def a_upload():
    while not stop:
        A.upload(**b0.arrays); A.sync()
def a_upload_run():
    while not stop:
        A.upload(**b0.arrays); A.run(); A.sync()
def measure(what, n=40):
    t0 = time.perf_counter()
    for _ in range(n):
        B.run(); B.sync()
    print("%-44s B replay %.3f ms/step" % (what, (time.perf_counter() - t0) / n * 1e3))
def measure_upload(what, n=20):
    t0 = time.perf_counter()
    for _ in range(n):
        B.upload(**b1.arrays); B.sync()
    print("%-44s B upload+prep %.3f ms" % (what, (time.perf_counter() - t0) / n * 1e3))
measure("B alone"); measure_upload("B alone")
for fn, name in ((a_copy_only, "A: results only (FSEG_DEBUG_RECOPY=1)"), (a_run_only, "A: replay loop"), (a_results, "A: run + results (D2H 75 MB) loop"), (a_upload, "A: upload + prep loop"), (a_upload_run, "A: upload + first run loop")):
    stop = False
    t = threading.Thread(target=fn); t.start(); time.sleep(0.05)
    measure("beside " + name); measure_upload("beside " + name)
    stop = True; t.join()
A.close(); B.close()
