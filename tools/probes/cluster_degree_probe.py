#!/usr/bin/env python3
"""Degree statistics of the big N3 tint's compatibility graph (before / after pruning): what k_prune's neighbour walks are made of."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cluster_util as cu
from freddie_amd import cluster_prep
t = cu.random_tint(1000, 20000, 2000, n_isoforms=8)
u = cluster_prep.unique_structures(t)
packed = cluster_prep.pack_structures([u])
ctx = cluster_prep.Context(0)
for prune in (False, True):
    adj, rounds = ctx.compat_graph(packed, prune=prune)
    n = len(u); aw = (n + 63) // 64
    a = np.asarray(adj[:n * aw], np.uint64).reshape(n, aw)
    deg = np.bitwise_count(a).sum(1)
    print("prune=%s rounds=%s n=%d: degree mean %.1f median %d max %d, isolated %d, deg1 %d, edges %d" % (prune, rounds.tolist(), n, deg.mean(), np.median(deg), deg.max(), int((deg == 0).sum()), int((deg == 1).sum()), int(deg.sum()) // 2))
ctx.close()
