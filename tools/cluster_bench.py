#!/usr/bin/env python3
"""Measurement of row N3 (clustering pre-ILP graph work) on MI355X: pairwise compatibility + pruning of a batch of
synthetic preprocessed tints through include/freddie_cluster.h.

    python bench.py --workload cluster-many|cluster-big [--steps K] [--no-cpu-baseline]     (with the CPU baseline)
    python tools/cluster_bench.py [--workload many|big] [--steps K]                          (GPU side only)

One JSON line: read pairs tested per second of the CALL (packed host arrays in -> pruned adjacency in host memory: copies,
kernels and the pruning loop's host round trips all inside), the kernel times as detail, and the bound of the compatibility
kernel.  That kernel works on bit rows staged in LDS, so bytes do not describe it (a byte roofline credited it with more
than the bus can move); its bound is integer VALU work: per pair and 32-segment word of the pair's overlap 2 logic ops +
2 popcounts + 2 adds (py/freddie_cluster.py:229,232 as bit rows), against 256 CUs x 64 lanes x 2.4 GHz lane-ops/s.  The CPU baseline (the oracle's Python
restatement on a bounded sample of the same tints) is bench.py's leg: this file never touches oracle/.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cluster_util as cu  # noqa: E402
from freddie_amd import cluster_prep  # noqa: E402

WORKLOADS = {"many": dict(n_tints=400, n_reps=500, n_segs=300), "big": dict(n_tints=1, n_reps=20000, n_segs=2000)}


def overlap_words(packed):
    """sum over unordered pairs with a non-empty overlap [max f, min l] of the 32-segment words that overlap spans."""
    total = 0
    for t in range(packed["n_tint"]):
        a, b = int(packed["row_off"][t]), int(packed["row_off"][t + 1])
        f = packed["first"][a:b].astype(np.int64); l = packed["last"][a:b].astype(np.int64)
        for i0 in range(0, b - a, 2048):
            lo = np.maximum(f[i0:i0 + 2048, None], f[None, :]); hi = np.minimum(l[i0:i0 + 2048, None], l[None, :])
            w = (hi >> 5) - (lo >> 5) + 1
            total += int(w[(hi >= lo) & (lo >= 0)].sum())
        own = (l >> 5) - (f >> 5) + 1
        total -= int(own[(l >= f) & (f >= 0)].sum())              # the diagonal
    return total // 2


def run(workload="many", steps=5, cpu_baseline=None):
    """Returns the result dict.  cpu_baseline: a callable(unique structures per tint) -> dict supplied by bench.py (the
    only place besides the tests that may use the oracle), or None."""
    w = WORKLOADS[workload]
    tints = [cu.random_tint(1000 + t, w["n_reps"], w["n_segs"], n_isoforms=8) for t in range(w["n_tints"])]
    uniq = [cluster_prep.unique_structures(t) for t in tints]
    packed = cluster_prep.pack_structures(uniq)
    n_pairs = sum(len(u) * (len(u) - 1) // 2 for u in uniq)
    words = overlap_words(packed)
    ctx = cluster_prep.Context(0)
    ctx.compat_graph(packed)                                     # warm-up
    compat, prune, wall = [], [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        adj, rounds = ctx.compat_graph(packed)
        wall.append(time.perf_counter() - t0)
        tm = ctx.last_timing()
        compat.append(tm["compat_ms"]); prune.append(tm["prune_ms"])
    ctx.close()
    compat_ms, prune_ms = float(np.mean(compat)), float(np.mean(prune))
    wall_ms = float(np.median(wall)) * 1e3
    lane_ops = 6 * words
    peak = 256 * 64 * 2.4                                          # G lane-ops/s
    out = {
        "metric": "read pairs tested/sec (compatibility graph + pruning, whole call: host arrays in, adjacency out)",
        "value": n_pairs / (wall_ms * 1e-3),
        "unit": "pairs/s", "n_gpus": 1, "steps": steps, "higher_is_better": True, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "cluster-" + workload, **w, "unique_reads": int(packed["row_off"][-1]),
                   "pairs": n_pairs, "prune_passes_max": int(rounds.max())},
        "call_wall_ms": wall_ms, "kernel_ms": {"compat": compat_ms, "prune": prune_ms},
        "value_kernels_only": n_pairs / ((compat_ms + prune_ms) * 1e-3),
        "roofline": {"kernel": "k_compat", "bound": "valu (integer: and / xor / popcount over bit rows in LDS)",
                     "achieved": lane_ops / (compat_ms * 1e-3) / 1e9, "peak": peak, "unit": "G lane-op/s",
                     "frac": lane_ops / (compat_ms * 1e-3) / 1e9 / peak, "traffic": None,
                     "algorithmic_lane_ops_per_launch": lane_ops, "overlap_words": words},
    }
    if cpu_baseline is not None:
        out["cpu_baseline"] = cpu_baseline(uniq)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="many", choices=sorted(WORKLOADS))
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    print(json.dumps(run(args.workload, args.steps)))


if __name__ == "__main__":
    main()
