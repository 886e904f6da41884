"""Where a fresh process's first 0.2 s go (the CLI's fixed costs, round 6): library load, the first context (hipInit), the second, the first run
of a small batch on each (code objects are loaded at a kernel's first launch), the second run.
    python tools/probes/startup_probe.py [n_partitions]"""
import os, sys, time
t_start = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
t_np = time.perf_counter()
from freddie_amd import _lib, tables, synth, pack
t_imp = time.perf_counter()
L = _lib.load()
t_load = time.perf_counter()
n_part = int(sys.argv[1]) if len(sys.argv) > 1 else 124
w = dict(synth.WORKLOADS["config4"]); w.pop("n_partitions")
parts = []
for i in range(n_part):
    g = synth.generate(i, with_seq=False, **w)
    parts.append(pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))
import bench
b = bench.Batch(parts)
params = bench.PARAMS["default"]
tabs = dict(w_main=tables.gaussian_half_kernel(params["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(params["sigma"], 1.0),
            h_table=np.asarray(tables.smooth_threshold(params["threshold_rate"]), np.float64))
print("numpy %.3f s, package %.3f s, dlopen %.3f s; batch of %d reads made" % (t_np - t_start, t_imp - t_np, t_load - t_imp, b.n_reads))
t0 = time.perf_counter(); c0 = _lib.Context(0); t1 = time.perf_counter(); c1 = _lib.Context(0); t2 = time.perf_counter()
print("first context %.3f s (hipInit inside), second %.3f s" % (t1 - t0, t2 - t1))
for name, c in (("context 0", c0), ("context 1", c1)):
    t0 = time.perf_counter(); c.set_params(**params, **tabs); t1 = time.perf_counter()
    c.upload(**b.arrays); t2 = time.perf_counter(); c.run(); c.sync(); t3 = time.perf_counter(); r = c.results(packed=True); t4 = time.perf_counter()
    c.upload(**b.arrays); t5 = time.perf_counter(); c.run(); c.sync(); t6 = time.perf_counter(); r = c.results(packed=True); t7 = time.perf_counter()
    print("%s: set_params %.1f ms | first: upload %.1f, run %.1f, results %.1f ms | second: upload %.1f, run %.1f, results %.1f ms" % (
        name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3, (t7 - t6) * 1e3))
c0.close(); c1.close()
