import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, goldens, util
from freddie_amd import _lib
g = goldens.load("g_refine"); part = goldens.partition_of(g)
ctx = _lib.Context(0)
util.run_gpu(ctx, [part], goldens.params_of(g), goldens.tables_of(g))
fy = ctx.tap("final_y"); ref = goldens.as_oracle_result(g)
print("gpu finals", len(fy), "ref", len(ref["final_y"]))
a=set(fy.tolist()); b=set(ref["final_y"].tolist())
print("missing", sorted(b-a)[:20], "extra", sorted(a-b)[:20])
print("chosen ok", np.array_equal(ctx.tap("chosen"), np.isin(np.arange(len(g["cands"])), g["finalc"]).astype(np.uint8)))
