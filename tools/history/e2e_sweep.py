#!/usr/bin/env python3
"""Wall time of the drop-in CLI on one split directory against its host-side knobs: python tools/e2e_sweep.py <split dir> [reads]"""
import os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
split = sys.argv[1]
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
out = os.path.join(os.path.dirname(split.rstrip("/")), "out_sweep")
for br in (250000, 125000, 62500):
    for t in (8, 16, 24):
        best = 1e9
        for _ in range(2):
            shutil.rmtree(out, ignore_errors=True)
            t0 = time.perf_counter()
            subprocess.check_call([sys.executable, os.path.join(ROOT, "py", "freddie_segment.py"), "-s", split, "-o", out, "-t", str(t),
                                   "--gpus", "1", "--sidecar", "off", "--batch-reads", str(br)], stdout=subprocess.DEVNULL)
            best = min(best, time.perf_counter() - t0)
        print("--batch-reads %6d -t %2d: %.2f s  %.2f M reads/s" % (br, t, best, n_reads / best / 1e6), flush=True)
shutil.rmtree(out, ignore_errors=True)
