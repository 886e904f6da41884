#!/usr/bin/env python3
"""Measurement of row N4 (isoform consensus counts + boundary votes) on MI355X through include/freddie_isoforms.h.

    python bench.py --workload isoforms [--steps K] [--no-cpu-baseline]                                (with the CPU baseline)
    python tools/isoforms_bench.py [--isoforms N] [--reads-per-isoform R] [--segments M] [--steps K]     (GPU side only)

One JSON line: reads processed per second of the two CALLS (host arrays in -> counts / votes in host memory; the 300 MB of
labels crossing PCIe dominate it), the kernel times as detail, and the consensus kernel's roofline with ALGORITHMIC bytes =
one label byte per (read, segment) + two int32 per (isoform, segment) out -- what isoforms_cons() reads and produces once
(py/freddie_isoforms.py:203-232).  The CPU baseline (the Python oracle on a bounded sample) is bench.py's leg: this file
never touches oracle/.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freddie_amd import isoforms  # noqa: E402


def run(n_iso=4000, per=500, M=150, steps=5, cpu_baseline=None):
    """Returns the result dict.  cpu_baseline: a callable(reads per isoform, segments, window) -> dict supplied by
    bench.py (the only place besides the tests that may use the oracle), or None."""
    rng = np.random.default_rng(11)
    R = n_iso * per
    lab = rng.choice(np.frombuffer(b"0012", np.uint8), size=(R, M), p=[0.3, 0.3, 0.3, 0.1])
    tail = rng.integers(0, 3, R).astype(np.uint8)
    iro = np.arange(n_iso + 1, dtype=np.int64) * per
    nb, w = 12, 8
    iso_b = np.sort(rng.integers(0, 50000, (n_iso, nb)), axis=1).astype(np.int32)
    rb = rng.integers(0, 50000, (R, 8)).astype(np.int32)
    ctx = isoforms.Context(0)
    ctx.consensus(iro, np.full(n_iso, M), np.arange(R, dtype=np.int64) * M, lab.reshape(-1), tail)
    cons_ms, votes_ms, wall = [], [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        ctx.kernel_ms = 0.0
        ctx.consensus(iro, np.full(n_iso, M), np.arange(R, dtype=np.int64) * M, lab.reshape(-1), tail)
        cons_ms.append(ctx.kernel_ms); ctx.kernel_ms = 0.0
        ctx.boundary_votes(iro, np.arange(n_iso + 1, dtype=np.int64) * nb, iso_b.reshape(-1), np.arange(R + 1, dtype=np.int64) * 8,
                           rb.reshape(-1), w)
        votes_ms.append(ctx.kernel_ms)
        wall.append(time.perf_counter() - t0)
    # the same counts from labels at two bits each (fiso_consensus_packed): what a producer that holds the packed form pays
    lab2 = isoforms.pack_labels(lab.reshape(-1))
    ref = ctx.consensus(iro, np.full(n_iso, M), np.arange(R, dtype=np.int64) * M, lab.reshape(-1), tail)
    got = ctx.consensus(iro, np.full(n_iso, M), np.arange(R, dtype=np.int64) * M, lab2, tail, packed=True)
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    pk_ms, pk_wall = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        ctx.kernel_ms = 0.0
        ctx.consensus(iro, np.full(n_iso, M), np.arange(R, dtype=np.int64) * M, lab2, tail, packed=True)
        pk_ms.append(ctx.kernel_ms)
        pk_wall.append(time.perf_counter() - t0)
    ctx.close()
    c_ms, v_ms = float(np.mean(cons_ms)), float(np.mean(votes_ms))
    alg = R * M + 8 * n_iso * M
    wall_ms = float(np.median(wall)) * 1e3
    out = {"metric": "reads/sec (isoform consensus counts + boundary votes, whole calls: host arrays in, results out)",
           "value": R / (wall_ms * 1e-3), "unit": "reads/s", "value_kernels_only": R / ((c_ms + v_ms) * 1e-3),
           "n_gpus": 1, "steps": steps, "higher_is_better": True, "dtype": "u8/int32", "data": "synthetic",
           "config": {"workload": "isoforms", "isoforms": n_iso, "reads": R, "segments": M, "window": w},
           "kernel_ms": {"consensus": c_ms, "votes": v_ms}, "call_wall_ms": wall_ms,
           "roofline": {"kernel": "k_consensus_rows (rows of at most 1 024 labels: one pass; longer rows: the two-pass k_consensus)", "bound": "hbm", "achieved": alg / (c_ms * 1e-3) / 1e9, "peak": 8000.0,
                        "unit": "GB/s", "frac": alg / (c_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "algorithmic_bytes_per_launch": alg}}
    p_ms = float(np.mean(pk_ms))
    out["consensus_packed_labels"] = {"kernel_ms": p_ms, "call_wall_ms": float(np.median(pk_wall)) * 1e3,
                                      "roofline_frac": alg / (p_ms * 1e-3) / 1e9 / 8000.0,
                                      "what": "fiso_consensus_packed: the consensus call alone with the labels at two bits each (75 MB "
                                              "instead of 300 MB across PCIe); algorithmic bytes as above (one byte per label)"}
    if cpu_baseline is not None:
        out["cpu_baseline"] = cpu_baseline(per, M, w)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--isoforms", type=int, default=4000)
    ap.add_argument("--reads-per-isoform", type=int, default=500)
    ap.add_argument("--segments", type=int, default=150)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    print(json.dumps(run(a.isoforms, a.reads_per_isoform, a.segments, a.steps)))


if __name__ == "__main__":
    main()
