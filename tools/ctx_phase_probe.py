#!/usr/bin/env python3
"""Where a context's host thread spends a batch in the timed steps of bench.py (config4, N contexts on one GPU).

Every thread does what bench.one_shot_steps does -- upload, run, results(packed) -- with a wall clock and the thread's CPU
clock around each call; variants leave a phase out (results; upload: the batch stays resident and is run again) so that what
the copies cost the job shows as a rate.  GPU box only (tools/host_ceiling.py is the host-only half).

    python tools/ctx_phase_probe.py [--contexts 8] [--steps 20]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from freddie_amd import tables  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contexts", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--workload", default="config4")
    args = ap.parse_args()
    params = bench.PARAMS["config5" if args.workload == "config5" else "default"]
    tabs = dict(w_main=tables.gaussian_half_kernel(params["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(params["sigma"], 1.0),
                h_table=np.asarray(tables.smooth_threshold(params["threshold_rate"]), np.float64))
    batches = bench.build_batches(args.workload, 0, 1)
    n_b = len(batches)
    import torch
    from freddie_amd import _lib
    torch.cuda.set_device(0)
    ctxs = [_lib.Context(0) for _ in range(args.contexts)]
    for c in ctxs:
        c.set_params(**params, **tabs)
        c.set_profiling(2)
    bench.one_shot_steps(ctxs, batches, [i % n_b for i in range(4 * n_b)])
    reads = batches[0].n_reads

    def leg(name, do_upload, do_results, packed=True):
        order = [i % n_b for i in range(args.steps * n_b)]
        acc = np.zeros((len(ctxs), 6))
        if not do_upload:
            for k, c in enumerate(ctxs):
                c.upload(**batches[k % n_b].arrays); c.run(); c.sync()

        def worker(k):
            c = ctxs[k]
            for si in range(k, len(order), len(ctxs)):
                b = batches[order[si]]
                w0, p0 = time.perf_counter(), time.thread_time()
                if do_upload:
                    c.upload(**b.arrays)
                w1, p1 = time.perf_counter(), time.thread_time()
                c.run()
                if not do_results:
                    c.sync()
                w2, p2 = time.perf_counter(), time.thread_time()
                if do_results:
                    c.results(packed=packed)
                w3, p3 = time.perf_counter(), time.thread_time()
                acc[k] += (w1 - w0, w2 - w1, w3 - w2, p1 - p0, p2 - p1, p3 - p2)
        th = [threading.Thread(target=worker, args=(k,)) for k in range(len(ctxs))]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = len(order)
        per = acc.sum(axis=0) / n * 1e3
        print("%-34s %7.1f M reads/s  %.3f ms/batch | per batch, one thread (wall / cpu ms): upload %.2f / %.2f  run %.2f / %.2f  results %.2f / %.2f"
              % (name, reads * n / dt / 1e6, dt / n * 1e3, per[0], per[3], per[1], per[4], per[2], per[5]), flush=True)

    for rep in range(2):
        leg("upload + run + results(packed)", True, True)
        leg("upload + run (+ sync)", True, False)
        leg("run + results(packed), resident", False, True)
        leg("run (+ sync), resident", False, False)


if __name__ == "__main__":
    main()
