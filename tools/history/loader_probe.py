#!/usr/bin/env python3
"""Raw speed of the two loaders of the native host library on this machine (no GPU work):
    python tools/loader_probe.py [--partitions N] [--reads R] [--threads T]"""
import argparse, glob, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freddie_amd import synth, _host
ap = argparse.ArgumentParser()
ap.add_argument("--partitions", type=int, default=1000)
ap.add_argument("--reads", type=int, default=500)
ap.add_argument("--threads", type=int, default=8)
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="loader_")
try:
    for i in range(a.partitions):
        synth.generate(i, n_reads=a.reads, n_exons=150, rp=0.05, write_dir=d)
    sp = sorted(glob.glob(d + "/chrS/split_*.tsv")); rp = [p.replace("split_", "reads_") for p in sp]; sc = [p[:-4] + ".fsc" for p in sp]
    hb = _host.HostBatch(sp, rp, n_threads=a.threads); hb.write_sidecars(sc, n_threads=a.threads); hb.close()
    for rep in range(3):
        for name, kw in (("tsv", {}), ("fsc", dict(sidecar_paths=sc)), ("fsc, no checksum", dict(sidecar_paths=sc, verify_checksum=False))):
            t = time.perf_counter(); hb = _host.HostBatch(sp, rp, n_threads=a.threads, **kw); dt = time.perf_counter() - t
            assert hb.n_from_sidecar == (0 if name == "tsv" else a.partitions)
            hb.close()
            print("%-18s %.3f s for %d reads (%d threads)" % (name, dt, a.partitions * a.reads, a.threads))
finally:
    shutil.rmtree(d, ignore_errors=True)
