#!/usr/bin/env python3
"""Emit the binary side-cars (split_<contig>_<tint>.fsc) of a Freddie split directory, so that the segmentation
stage loads flat arrays instead of parsing split_*.tsv / reads_*.tsv again.

    freddie_sidecar.py -s SPLIT_DIR [-t THREADS]

Run it once after freddie_split.py (or let `freddie_segment.py --sidecar write` do it on its first pass).  A side-car
records the sizes and mtimes of its two TSVs and is ignored as soon as they change; the TSVs stay the contract."""
import argparse
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from freddie_amd import _host  # noqa: E402
from freddie_amd.segment import sidecar_path  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description="Write binary side-cars next to the split TSVs")
    ap.add_argument("-s", "--split-dir", required=True)
    ap.add_argument("-t", "--threads", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=256, help="Partitions parsed per native call")
    args = ap.parse_args(argv)
    split_dir = args.split_dir.rstrip("/")
    splits = sorted(glob.glob("{}/*/split_*.tsv".format(split_dir)))
    done = 0
    for i in range(0, len(splits), args.chunk):
        sp = splits[i:i + args.chunk]
        rp = [os.path.join(os.path.dirname(p), "reads_" + os.path.basename(p)[len("split_"):]) for p in sp]
        scs = [sidecar_path(p) for p in sp]
        hb = _host.HostBatch(sp, rp, n_threads=args.threads, sidecar_paths=scs)
        try:
            if hb.n_from_sidecar < hb.n_part:
                hb.write_sidecars(scs, n_threads=args.threads)
            done += hb.n_part
        finally:
            hb.close()
    print("[freddie_sidecar] {} partitions have fresh side-cars".format(done))


if __name__ == "__main__":
    main()
