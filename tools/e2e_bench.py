#!/usr/bin/env python3
"""End-to-end wall time of the drop-in CLI on a synthetic split directory (files in, files out).
    python tools/e2e_bench.py [--partitions N] [--reads R] [--threads T] [--keep DIR]
Prints reads/s including parsing, upload, GPU, download, annotation and writing.  Not the bench.py metric
(that one starts with inputs resident in HBM); DESIGN.md quotes both."""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freddie_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--partitions", type=int, default=200)
ap.add_argument("--reads", type=int, default=500)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--keep", default=None)
ap.add_argument("--generate-only", action="store_true")
ap.add_argument("--sidecar", default="off", choices=("auto", "off", "write"))
ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("--timing", action="store_true", help="per-batch load / device / write seconds on stderr")
args = ap.parse_args()
work = args.keep or tempfile.mkdtemp(prefix="e2e_")
split = os.path.join(work, "split")
if not os.path.isdir(split):
    t0 = time.time()
    for i in range(args.partitions):
        synth.generate(i, n_reads=args.reads, n_exons=150, rp=0.05, write_dir=split)
    print("generated %d partitions in %.1f s" % (args.partitions, time.time() - t0))
if args.generate_only:
    sys.exit(0)
out = os.path.join(work, "out")
n = args.partitions * args.reads
for rep in range(args.repeat):
    shutil.rmtree(out, ignore_errors=True)
    t0 = time.time()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "py", "freddie_segment.py"), "-s", split, "-o", out,
                           "-t", str(args.threads), "--gpus", "1", "--sidecar", args.sidecar], stdout=subprocess.DEVNULL,
                          env=dict(os.environ, FREDDIE_TIMING="1" if args.timing else "0"))
    dt = time.time() - t0
    size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(split) for f in fs if f.endswith(".tsv"))
    fsc = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(split) for f in fs if f.endswith(".fsc"))
    print("e2e[--sidecar %s, run %d]: %d reads in %d partitions, %.2f s wall (incl. interpreter + context start-up), "
          "%.0f reads/s, TSV input %.1f MB, side-cars %.1f MB" % (args.sidecar, rep, n, args.partitions, dt, n / dt,
                                                                   size / 1e6, fsc / 1e6))
if not args.keep:
    shutil.rmtree(work, ignore_errors=True)
