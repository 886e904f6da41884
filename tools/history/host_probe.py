#!/usr/bin/env python3
"""Host-side throughput of the native loader and writer (libfreddie_host.so), no GPU: a synthetic split directory in
tmpfs is parsed with T threads, then annotated and written from results the CPU oracle produced once (cached).

    python tools/host_probe.py [--partitions 200] [--reads 500] [--threads 8] [--repeat 3] [--dir /dev/shm/freddie_hostprobe]
The oracle is used only to fabricate realistic device results for the writer (this is a developer tool, not the product)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from freddie_amd import _host, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--partitions", type=int, default=200)
ap.add_argument("--reads", type=int, default=500)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--repeat", type=int, default=3)
ap.add_argument("--dir", default="/dev/shm/freddie_hostprobe")
ap.add_argument("--sidecar", action="store_true")
args = ap.parse_args()

split = os.path.join(args.dir, "split_%d_%d" % (args.partitions, args.reads))
if not os.path.isdir(split):
    for i in range(args.partitions):
        synth.generate(i, n_reads=args.reads, n_exons=150, rp=0.05, write_dir=split)
sp = [os.path.join(split, "chrS", "split_chrS_%d.tsv" % i) for i in range(args.partitions)]
rp = [os.path.join(split, "chrS", "reads_chrS_%d.tsv" % i) for i in range(args.partitions)]
sc = [p[:-4] + ".fsc" for p in sp]
size = sum(os.path.getsize(p) for p in sp + rp)
res_path = os.path.join(split, "results.npz")
hb = _host.HostBatch(sp, rp, n_threads=args.threads)
if args.sidecar and not all(os.path.exists(p) for p in sc):
    hb.write_sidecars(sc, n_threads=args.threads)
if not os.path.exists(res_path):
    import util  # tests/util.py: oracle runner
    from freddie_amd import pack
    a = hb.arrays()
    pfo, lo, fps, labs = [0], [0], [], []
    for p in range(hb.n_part):
        k0, k1 = a["part_iv_off"][p], a["part_iv_off"][p + 1]
        r0, r1 = a["part_rep_off"][p], a["part_rep_off"][p + 1]
        e0, e1 = a["rep_exon_off"][r0], a["rep_exon_off"][r1]
        part = pack.PackedPartition(a["iv_start"][k0:k1], a["iv_end"][k0:k1], a["rep_weight"][r0:r1],
                                    a["rep_exon_off"][r0:r1 + 1] - e0, a["ex_ts"][e0:e1], a["ex_te"][e0:e1], np.zeros(0, np.int32))
        o = util.run_oracle(part)
        assert not o["error"], o["errmsg"]
        fps.append(o["final_pos"]); labs.append((o["labels"] + 48).astype(np.uint8).ravel())
        pfo.append(pfo[-1] + len(o["final_pos"])); lo.append(lo[-1] + labs[-1].size)
    np.savez(res_path, pfo=np.array(pfo), fp=np.concatenate(fps), lo=np.array(lo), lab=np.concatenate(labs))
hb.close()
r = np.load(res_path)
out_dir = os.path.join(args.dir, "out")
os.makedirs(out_dir, exist_ok=True)
outs = [os.path.join(out_dir, "segment_%d.tsv" % i) for i in range(args.partitions)]
n_reads = args.partitions * args.reads
for rep in range(args.repeat):
    t0 = time.perf_counter()
    hb = _host.HostBatch(sp, rp, n_threads=args.threads, sidecar_paths=sc if args.sidecar else None)
    t1 = time.perf_counter()
    hb.write(r["pfo"], r["fp"], r["lo"], r["lab"], outs, n_threads=args.threads)
    t2 = time.perf_counter()
    hb.close()
    t3 = time.perf_counter()
    osz = sum(os.path.getsize(p) for p in outs)
    print("run %d (-t %d%s): load %.3f s (%.0f MB/s, %.2f M reads/s), write %.3f s (%.2f M reads/s, %.0f MB out), free %.3f s"
          % (rep, args.threads, ", side-cars" if args.sidecar else "", t1 - t0, size / 1e6 / (t1 - t0), n_reads / 1e6 / (t1 - t0),
             t2 - t1, n_reads / 1e6 / (t2 - t1), osz / 1e6, t3 - t2))
