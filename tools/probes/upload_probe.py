import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import bench
from freddie_amd import tables, _lib
params = bench.PARAMS["default"]
tabs = dict(w_main=tables.gaussian_half_kernel(params["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(params["sigma"], 1.0),
            h_table=np.asarray(tables.smooth_threshold(params["threshold_rate"]), np.float64))
batches = bench.build_batches("config4", 0, 1)
import torch
torch.cuda.set_device(0)
c = _lib.Context(0); c.set_params(**params, **tabs)
for i in range(4):
    b = batches[i % len(batches)]
    t0 = time.perf_counter(); c.upload(**b.arrays); t1 = time.perf_counter(); c.run(); t2 = time.perf_counter(); r = c.results(packed=True); t3 = time.perf_counter()
    print("upload %.3f run %.3f results %.3f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), {k: v.nbytes for k, v in b.arrays.items()})
