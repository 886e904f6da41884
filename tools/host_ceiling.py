#!/usr/bin/env python3
"""How many reads per second the HOST side of the drop-in CLI can take, without any GPU: the real main() (discovery, LPT
scatter, worker processes, loader / device / writer threads, native parser and writer: freddie_amd/segment.py, reference
main() py/freddie_segment.py:847-885) with the device replaced by a stand-in context that answers at once -- every tint
interval's first and last position as the final positions and a constant label for every (read rep, segment).

    python tools/host_ceiling.py [--partitions 4000] [--reads 500] [--workers 1,2,4,8] [--threads T] [--label 0|1] [--keep DIR]

--label 0: no read has a '1' label, so the writer's gaps / poly-A annotation has nothing to do (py/freddie_segment.py:371-373):
the floor of the write cost.  --label 1: every segment of every read is '1', so every read pays the CIGAR walks and both
poly-A scans: the ceiling.  Real outputs lie in between (a read covers a tenth of its partition's segments).
What the numbers mean: with G GPU workers the CLI cannot go faster than this whatever the GPUs do; the device pipeline takes
~300 M reads/s per GPU (bench.py), so this IS the end-to-end ceiling of a node."""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freddie_amd import segment, synth  # noqa: E402


class NullContext:
    """Same calls as _lib.Context; results of the right shape, no work."""
    label = 0

    def __init__(self, device):
        self.device = device

    def set_params(self, *a, **kw):
        pass

    def upload(self, part_iv_off, iv_start, iv_end, part_rep_off, rep_weight, rep_exon_off, ex_ts, ex_te):
        pio = np.asarray(part_iv_off, np.int64)
        pro = np.asarray(part_rep_off, np.int64)
        K = np.diff(pio)
        F = 2 * K                                              # final positions of a partition: both ends of every interval
        pfo = np.zeros(len(K) + 1, np.int64)
        np.cumsum(F, out=pfo[1:])
        fp = np.empty(int(pfo[-1]), np.int32)
        fp[0::2] = np.asarray(iv_start, np.int32)
        fp[1::2] = np.asarray(iv_end, np.int32)
        lo = np.zeros(len(K) + 1, np.int64)
        np.cumsum(np.diff(pro) * (F - 1), out=lo[1:])
        packed = np.full(int((lo[-1] + 15) // 16 * 4), 0x55 if self.label else 0, np.uint8)   # two bits per label
        self.res = (pfo, fp, lo, packed)

    def run(self):
        pass

    def results(self, packed=False):
        assert packed, "the CLI fetches packed labels"
        return self.res

    def close(self):
        pass


def null_contexts(device, n=2):
    return [NullContext(device) for _ in range(n)]


def generate(split, n_part, n_reads, n_contigs=1):
    """Partitions 0 .. n_part - 1, spread over n_contigs contig directories (chr1 .. chrN; one: chrS, as the bench's e2e leg) --
    a genome's split directory has a directory per contig (reference main() :852-857), and a directory is also what the
    output files of many partitions contend for."""
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(len(os.sched_getaffinity(0)), 32)) as pool:
        pool.map(_gen, [(i, n_reads, split, "chrS" if n_contigs <= 1 else "chr%d" % (1 + i % n_contigs)) for i in range(n_part)], chunksize=8)


def _gen(job):
    i, n_reads, split, contig = job
    synth.generate(i, n_reads=n_reads, n_exons=150, rp=0.05, write_dir=split, contig=contig)


def run(split, out, workers, threads, label):
    NullContext.label = label
    segment.open_contexts = null_contexts
    segment.WORKER_START_METHOD = "fork"                       # the stand-in has to reach the worker processes
    shutil.rmtree(out, ignore_errors=True)
    t0 = time.perf_counter()
    stdout = sys.stdout
    sys.stdout = open(os.devnull, "w")                         # the reference's progress lines
    try:
        segment.main(["-s", split, "-o", out, "-t", str(threads), "--devices", ",".join(str(w) for w in range(workers)), "--sidecar", "off"])
    finally:
        sys.stdout.close()
        sys.stdout = stdout
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--partitions", type=int, default=4000)
    ap.add_argument("--reads", type=int, default=500)
    ap.add_argument("--workers", default="1,2,4,8")
    ap.add_argument("--threads", type=int, default=0, help="-t per worker (default: host cores / workers, at most 16)")
    ap.add_argument("--label", default="0,1")
    ap.add_argument("--keep", default=None)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--contigs", type=int, default=1, help="contig directories the partitions are spread over (a genome: 24 and more)")
    args = ap.parse_args()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    work = args.keep or tempfile.mkdtemp(prefix="host_ceiling_", dir=base)
    split, out = os.path.join(work, "split"), os.path.join(work, "out")
    try:
        if not os.path.isdir(split):
            t0 = time.perf_counter()
            generate(split, args.partitions, args.reads, args.contigs)
            print("generated %d partitions x %d reads in %d contig directories in %.1f s under %s" % (
                args.partitions, args.reads, max(1, args.contigs), time.perf_counter() - t0, work))
        n = args.partitions * args.reads
        cores = len(os.sched_getaffinity(0))
        print("host cores available: %d" % cores)
        for label in [int(x) for x in args.label.split(",")]:
            for w in [int(x) for x in args.workers.split(",")]:
                th = args.threads or max(1, min(16, cores // w))
                walls = [run(split, out, w, th, label) for _ in range(args.repeat)]
                print("label %d, %d worker(s) x -t %d: %.2f s (runs: %s) -> %.2f M reads/s parsed and written" % (
                    label, w, th, min(walls), ", ".join("%.2f" % x for x in walls), n / min(walls) / 1e6))
    finally:
        if not args.keep:
            shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
