import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
from freddie_amd import _lib, tables
params = bench.PARAMS["default"]
tabs = dict(w_main=tables.gaussian_half_kernel(params["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(params["sigma"], 1.0),
            h_table=np.asarray(tables.smooth_threshold(params["threshold_rate"]), np.float64))
w = dict(bench.synth.WORKLOADS["config4"]); n = w.pop("n_partitions")
parts = []
for i in range(n):
    g = bench.synth.generate(i, with_seq=False, **w)
    parts.append(bench.pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te, dedupe=True))
b = bench.Batch(parts)
torch.cuda.set_device(0)
f0, tot = torch.cuda.mem_get_info()
print("free %.1f GB of %.1f" % (f0 / 1e9, tot / 1e9))
ctxs = []
for k in range(8):
    c = _lib.Context(0); c.set_params(**params, **tabs); c.upload(**b.arrays); c.run(); c.sync(); c.results(packed=True); c.run(); c.sync()
    ctxs.append(c)
    f, _ = torch.cuda.mem_get_info()
    print("after context %d: used %.1f GB" % (k, (f0 - f) / 1e9))
for c in ctxs: c.close()
