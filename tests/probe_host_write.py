#!/usr/bin/env python3
"""Developer probe (not a test; no GPU): what the native writer (fhost_write_packed: gaps / poly-A annotation :370-472 + TSV assembly
:703-732) costs per read on REAL labels.  The labels come from the CPU oracle, which is why this lives under tests/.
    python tests/probe_host_write.py <work_dir_with_split/> [--parts 200] [--threads 1] [--repeat 5]
Environment FHOST_LIB=<other build> compares builds; FHOST_PROBE=1 builds print their own phase clocks."""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import util  # noqa: E402
from freddie_amd import _host, pack  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("work")
ap.add_argument("--parts", type=int, default=200)
ap.add_argument("--threads", type=int, default=1)
ap.add_argument("--repeat", type=int, default=5)
args = ap.parse_args()
split = os.path.join(args.work, "split")
jobs = []
for contig in sorted(os.listdir(split)):
    d = os.path.join(split, contig)
    for f in sorted(os.listdir(d)):
        if f.startswith("split_") and f.endswith(".tsv"):
            jobs.append((os.path.join(d, f), os.path.join(d, "reads_" + f[6:])))
jobs = jobs[:args.parts]
t0 = time.perf_counter()
hb = _host.HostBatch([j[0] for j in jobs], [j[1] for j in jobs], n_threads=args.threads)
t_load = time.perf_counter() - t0
a = hb.arrays()
pfo, fps, lo, labs = [0], [], [0], []
for p in range(hb.n_part):
    k0, k1 = a["part_iv_off"][p], a["part_iv_off"][p + 1]
    r0, r1 = a["part_rep_off"][p], a["part_rep_off"][p + 1]
    e0, e1 = a["rep_exon_off"][r0], a["rep_exon_off"][r1]
    part = pack.PackedPartition(a["iv_start"][k0:k1], a["iv_end"][k0:k1], a["rep_weight"][r0:r1], a["rep_exon_off"][r0:r1 + 1] - e0,
                                a["ex_ts"][e0:e1], a["ex_te"][e0:e1], None)
    o = util.run_oracle(part)
    assert o["error"] == 0, o["errmsg"]
    fps.append(o["final_pos"]); pfo.append(pfo[-1] + len(o["final_pos"]))
    labs.append((o["labels"] + 48).astype(np.uint8).ravel()); lo.append(lo[-1] + labs[-1].size)
labels = np.concatenate(labs)
packed = util.pack_labels(labels)
out_dir = os.path.join(args.work, "out_probe"); os.makedirs(out_dir, exist_ok=True)
outs = [os.path.join(out_dir, "seg_%d.tsv" % i) for i in range(hb.n_part)]
best = 1e9
for _ in range(args.repeat):
    t0 = time.perf_counter()
    hb.write(np.array(pfo, np.int64), np.concatenate(fps).astype(np.int32), np.array(lo, np.int64), packed, outs, n_threads=args.threads, packed=True)
    best = min(best, time.perf_counter() - t0)
size = sum(os.path.getsize(p) for p in outs)
print("%d partitions, %d reads: load %.3f s (%.2f us/read), write best of %d: %.3f s = %.2f us/read x %d thread(s); %.1f MB out, labels '1' share %.2f" % (
    hb.n_part, hb.n_reads, t_load, t_load / hb.n_reads * 1e6, args.repeat, best, best / hb.n_reads * 1e6 * args.threads, args.threads, size / 1e6,
    float((labels == 49).mean())))
hb.close()
