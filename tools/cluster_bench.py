#!/usr/bin/env python3
"""Measurement of row N3 (clustering pre-ILP graph work) on MI355X: pairwise compatibility + pruning of a batch of
synthetic preprocessed tints through include/freddie_cluster.h.

    python bench.py --workload cluster-many|cluster-big [--steps K] [--no-cpu-baseline]     (with the CPU baseline)
    python tools/cluster_bench.py [--workload many|big] [--steps K]                          (GPU side only)

One JSON line: read pairs tested per second (kernel time from HIP events on the library's stream, inputs already packed
on the host; the call's host->device copies are outside the event bracket), the roofline figure of the compatibility
kernel with ALGORITHMIC bytes = 2 label cells (1 B each) per segment of every pair's overlap -- what the reference's two
zip() passes over d1[f:l+1], d2[f:l+1] read once (py/freddie_cluster.py:229,232).  The CPU baseline (the oracle's Python
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


def overlap_cells(packed):
    """sum over unordered pairs of max(0, min(l) - max(f) + 1), per tint, exactly."""
    total = 0
    for t in range(packed["n_tint"]):
        a, b = int(packed["row_off"][t]), int(packed["row_off"][t + 1])
        f = packed["first"][a:b].astype(np.int64); l = packed["last"][a:b].astype(np.int64)
        for i0 in range(0, b - a, 2048):
            o = np.minimum(l[i0:i0 + 2048, None], l[None, :]) - np.maximum(f[i0:i0 + 2048, None], f[None, :]) + 1
            total += int(np.clip(o, 0, None).sum())
        total -= int(np.clip(l - f + 1, 0, None).sum())          # the diagonal
    return total // 2


def run(workload="many", steps=5, cpu_baseline=None):
    """Returns the result dict.  cpu_baseline: a callable(unique structures per tint) -> dict supplied by bench.py (the
    only place besides the tests that may use the oracle), or None."""
    w = WORKLOADS[workload]
    tints = [cu.random_tint(1000 + t, w["n_reps"], w["n_segs"], n_isoforms=8) for t in range(w["n_tints"])]
    uniq = [cluster_prep.unique_structures(t) for t in tints]
    packed = cluster_prep.pack_structures(uniq)
    n_pairs = sum(len(u) * (len(u) - 1) // 2 for u in uniq)
    cells = overlap_cells(packed)
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
    alg_bytes = 2 * cells
    out = {
        "metric": "read pairs tested/sec (compatibility graph + pruning, kernels)", "value": n_pairs / ((compat_ms + prune_ms) * 1e-3),
        "unit": "pairs/s", "n_gpus": 1, "steps": steps, "higher_is_better": True, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "cluster-" + workload, **w, "unique_reads": int(packed["row_off"][-1]),
                   "pairs": n_pairs, "prune_passes_max": int(rounds.max())},
        "kernel_ms": {"compat": compat_ms, "prune": prune_ms}, "call_wall_ms": float(np.mean(wall)) * 1e3,
        "roofline": {"kernel": "k_compat", "bound": "hbm", "achieved": alg_bytes / (compat_ms * 1e-3) / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": alg_bytes / (compat_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes},
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
