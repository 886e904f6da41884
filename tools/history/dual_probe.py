import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from freddie_amd import _lib, pack, tables
wl = sys.argv[1] if len(sys.argv) > 1 else "config4"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
params = bench.PARAMS["config5" if wl == "config5" else "default"]
tabs = dict(w_main=tables.gaussian_half_kernel(params["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(params["sigma"], 1.0),
            h_table=np.asarray(tables.smooth_threshold(params["threshold_rate"]), np.float64))
parts, n_reads = bench.build_batch(wl, 0, bench.per_gpu_partitions(wl, 1))
def mk(ps):
    c = _lib.Context(0); c.set_params(**params, **tabs); c.upload(**pack.concat_batch(ps)); return c
def timeit(ctxs, steps=60):
    for _ in range(5):
        for c in ctxs: c.run()
        for c in ctxs: c.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for c in ctxs: c.run()
        for c in ctxs: c.sync()
    return (time.perf_counter() - t0) / steps * 1e3
one = mk(parts)
print(wl, "single context: %.3f ms/step" % timeit([one]))
one.close()
chunks = [parts[i::S] for i in range(S)]
ctxs = [mk(ch) for ch in chunks]
print(wl, "%d contexts, interleaved partitions: %.3f ms/step" % (S, timeit(ctxs)))
for c in ctxs: c.close()
