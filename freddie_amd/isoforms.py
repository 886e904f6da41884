"""Host side of the isoform-consensus stage: the same interface as the reference's
``py/freddie_isoforms.py`` (CLI flags, ``cluster_*.tsv`` + ``split_*.tsv`` in, one GTF out), with the two per-read
loops -- consensus counts and boundary votes -- done by the gfx950 library behind ``include/freddie_isoforms.h`` for a
whole batch of tints per call.  There is no CPU implementation of those loops in this package.

Reference map (file:line of vpc-ccg/freddie ``py/freddie_isoforms.py``):
  parse_args :10-47 / main :253-287   -> parse_args(), main()
  read_cluster :159-200               -> read_cluster()
  read_split :143-156                 -> read_split()
  isoforms_cons :203-250              -> isoforms_cons_batch()      counts on the GPU, decisions here
  correct_boundaries :122-140         -> correct_boundaries_batch() votes on the GPU, decisions here
  get_gtf_records :72-119             -> get_gtf_records()
  run_consensus :50-69                -> run_consensus(), run_consensus_batch()
"""
import argparse
import ctypes
import glob
import os
from itertools import groupby

import numpy as np

from . import build as _build

ISO_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfreddie_isoforms.so")
ISO_SRC = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "freddie_isoforms.hip")]
EXPORTS = ["fiso_abi_version", "fiso_create", "fiso_destroy", "fiso_last_error", "fiso_consensus", "fiso_consensus_packed", "fiso_boundary_votes",
           "fiso_last_kernel_ms"]
_lib = None
_TAIL_CODE = {"N": 0, "S": 1, "E": 2}


class IsoformsError(RuntimeError):
    pass


def build(force=False, verbose=False):
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-I", _build.INCLUDE, "-o", ISO_SO] + ISO_SRC
    _build.build_stamped(ISO_SO, cmd, ISO_SRC + [os.path.join(_build.INCLUDE, "freddie_isoforms.h")], force, verbose)
    return ISO_SO


def load():
    global _lib
    if _lib is not None:
        return _lib
    so = os.environ.get("FISO_LIB") or ISO_SO          # (FISO_LIB: a variant build, tools/ only)
    if not os.path.exists(so):
        raise IsoformsError("%s not found: build it first (freddie_amd.isoforms.build()); there is no CPU fallback" % so)
    L = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    L.fiso_abi_version.restype = ctypes.c_int
    L.fiso_create.restype = ctypes.c_int
    L.fiso_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.fiso_destroy.restype = None
    L.fiso_destroy.argtypes = [vp]
    L.fiso_last_error.restype = ctypes.c_char_p
    L.fiso_last_error.argtypes = [vp]
    L.fiso_consensus.restype = ctypes.c_int
    L.fiso_consensus.argtypes = [vp, ctypes.c_int32] + [vp] * 9
    L.fiso_consensus_packed.restype = ctypes.c_int
    L.fiso_consensus_packed.argtypes = L.fiso_consensus.argtypes
    L.fiso_boundary_votes.restype = ctypes.c_int
    L.fiso_boundary_votes.argtypes = [vp, ctypes.c_int32, vp, vp, vp, vp, vp, ctypes.c_int32, vp]
    L.fiso_last_kernel_ms.restype = ctypes.c_int
    L.fiso_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    _lib = L
    return L


def pack_labels(labels):
    """ASCII labels '0' / '1' / '2' -> two bits each (label g at bits 2(g & 3).. of byte g >> 2)."""
    v = (np.ascontiguousarray(labels, np.uint8) & 3).astype(np.uint8)
    v = np.concatenate([v, np.zeros((-len(v)) % 4, np.uint8)]).reshape(-1, 4)
    return (v[:, 0] | (v[:, 1] << 2) | (v[:, 2] << 4) | (v[:, 3] << 6)).astype(np.uint8)


def _ptr(a):
    return a.ctypes.data if a.size else None


class Context:
    def __init__(self, device=0):
        self._L = load()
        h = ctypes.c_void_p()
        if self._L.fiso_create(int(device), ctypes.byref(h)) != 0:
            raise IsoformsError("fiso_create: " + self._L.fiso_last_error(None).decode())
        self._h = h
        self.kernel_ms = 0.0

    def _ms(self):
        v = ctypes.c_float()
        self._L.fiso_last_kernel_ms(self._h, ctypes.byref(v))
        self.kernel_ms += v.value

    def consensus(self, iso_read_off, n_seg, read_lab_off, labels, tail, packed=False):
        """packed: ``labels`` holds two bits per label (pack_labels() / the segmentation stage's packed results);
        read_lab_off counts labels either way."""
        n_iso = len(n_seg)
        a = [np.ascontiguousarray(iso_read_off, np.int64), np.ascontiguousarray(n_seg, np.int32), None,
             np.ascontiguousarray(read_lab_off, np.int64), np.ascontiguousarray(labels, np.uint8),
             np.ascontiguousarray(tail, np.uint8)]
        a[2] = np.zeros(n_iso + 1, np.int64)
        np.cumsum(a[1], out=a[2][1:])
        S = int(a[2][-1])
        cons = np.zeros(max(S, 1), np.int32); cov = np.zeros(max(S, 1), np.int32); tails = np.zeros(3 * n_iso, np.int32)
        rc = (self._L.fiso_consensus_packed if packed else self._L.fiso_consensus)(self._h, n_iso, _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), _ptr(a[4]), _ptr(a[5]),
                                    cons.ctypes.data, cov.ctypes.data, tails.ctypes.data)
        if rc != 0:
            raise IsoformsError("fiso_consensus: " + self._L.fiso_last_error(self._h).decode())
        self._ms()
        return a[2], cons[:S], cov[:S], tails.reshape(n_iso, 3)

    def boundary_votes(self, iso_read_off, iso_b_off, iso_bound, read_b_off, read_bound, window):
        a = [np.ascontiguousarray(iso_read_off, np.int64), np.ascontiguousarray(iso_b_off, np.int64),
             np.ascontiguousarray(iso_bound, np.int32), np.ascontiguousarray(read_b_off, np.int64),
             np.ascontiguousarray(read_bound, np.int32)]
        n_iso = len(a[0]) - 1
        votes = np.zeros((max(int(a[1][-1]), 1), 2 * window + 1), np.int32)
        rc = self._L.fiso_boundary_votes(self._h, n_iso, _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), _ptr(a[4]), int(window),
                                         votes.ctypes.data)
        if rc != 0:
            raise IsoformsError("fiso_boundary_votes: " + self._L.fiso_last_error(self._h).decode())
        self._ms()
        return votes[:int(a[1][-1])]

    def close(self):
        if getattr(self, "_h", None):
            self._L.fiso_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser(
        description="Extract alignment information from BAM/SAM file and splits reads into distinct transcriptional intervals")
    ap.add_argument("-s", "--split-dir", type=str, required=True, help="Path to directory of Freddie segment")
    ap.add_argument("-c", "--cluster-dir", type=str, required=True, help="Path to directory of Freddie cluster")
    ap.add_argument("-m", "--majority-threshold", type=float, default=0.50,
                    help="Majority threshold of reads to adjust exon boundary using the original alignments. Default: 0.5")
    ap.add_argument("-w", "--correction-window", type=int, default=8,
                    help="The +/- window around segment boundary to look for read alignment boundaries. Default: 8")
    ap.add_argument("-t", "--threads", type=int, default=1, help="Number of threads (kept for compatibility). Default: 1")
    ap.add_argument("-o", "--output", type=str, default="freddie_isoforms.gtf",
                    help="Path to output file. Default: freddie_isoforms.gtf")
    ap.add_argument("--device", type=int, default=0, help="GPU ordinal")
    ap.add_argument("--batch-tints", type=int, default=2000, help="Tints per device batch")
    args = ap.parse_args(argv)
    assert 0.5 <= args.majority_threshold <= 1.0
    assert 0 <= args.correction_window <= 20
    assert 0 < args.threads
    return args


def read_cluster(cluster_tsv):
    """(segments {(chrom, tint): [(s, e)]}, reads {rid: read}, isoforms {(chrom, tint, pid, iid): {'rids': set}});
    reads assigned to the garbage isoform ('*') and the isoform_ lines are skipped (:159-200)."""
    segments, reads, isoforms = dict(), dict(), dict()
    for line in open(cluster_tsv):
        f = line.rstrip().split("\t")
        if f[0][0] == "#":
            pos = [int(x) for x in f[2].split(",")]
            segments[(f[0][1:], int(f[1]))] = list(zip(pos[:-1], pos[1:]))
            continue
        if f[0].startswith("isoform_") or f[7] == "*":
            continue
        read = dict(rid=int(f[0]), rname=f[1], chrom=f[2], strand=f[3], tint=int(f[4]), pid=int(f[5]), tail=f[6],
                    iid=int(f[7]), data=f[8])
        assert len(read["data"]) == len(segments[(read["chrom"], read["tint"])])
        reads[read["rid"]] = read
        key = (read["chrom"], read["tint"], read["pid"], read["iid"])
        isoforms.setdefault(key, dict(rids=set()))["rids"].add(read["rid"])
    for isoform in isoforms.values():
        assert len({len(reads[rid]["data"]) for rid in isoform["rids"]}) == 1
    return segments, reads, isoforms


def read_split(split_tsv, reads):
    """starts / ends of the alignment intervals of the reads that were clustered (:143-156)."""
    for line in open(split_tsv):
        if line.startswith("#"):
            continue
        f = line.rstrip().split("\t")
        rid = int(f[0])
        if rid not in reads:
            continue
        pairs = [iv.split(":")[0].split("-") for iv in f[5:]]
        starts, ends = zip(*[(int(a), int(b)) for a, b in pairs])
        reads[rid]["starts"], reads[rid]["ends"] = starts, ends
        for s, e in zip(starts, ends):
            assert s < e


def isoforms_cons_batch(jobs, ctx):
    """isoforms_cons() (:203-250) of every (isoforms, segments, reads) job: one device call for the counts."""
    isos, iso_read_off, n_seg, lab, lab_off, tail = [], [0], [], [], [], []
    nbytes = 0
    for isoforms, segments, reads in jobs:
        for key, isoform in isoforms.items():
            M = len(segments[(key[0], key[1])])
            for rid in sorted(isoform["rids"]):
                read = reads[rid]
                assert len(read["data"]) == M, (M, key, read)
                lab.append(read["data"]); lab_off.append(nbytes); nbytes += M
                tail.append(_TAIL_CODE[read["tail"]])           # KeyError for an unknown category, as tails[...] += 1 (:231)
            isos.append((isoform, segments[(key[0], key[1])]))
            iso_read_off.append(len(lab)); n_seg.append(M)
    if not isos:
        return
    labels = np.frombuffer("".join(lab).encode("ascii"), np.uint8)
    seg_off, cons, cov, tails = ctx.consensus(iso_read_off, n_seg, lab_off, labels, tail)
    for i, (isoform, segs) in enumerate(isos):
        x = cons[seg_off[i]:seg_off[i + 1]].tolist(); c = cov[seg_off[i]:seg_off[i + 1]].tolist()
        flags = [a / b > 0.5 if a >= 3 else False for a, b in zip(x, c)]
        if True not in flags:
            continue
        isoform["strand"] = "-" if tails[i][1] > tails[i][2] else "+"
        starts, ends = [], []
        for d, group in groupby(enumerate(flags), lambda t: t[1]):
            if d is not True:
                continue
            group = list(group)
            starts.append(segs[group[0][0]][0]); ends.append(segs[group[-1][0]][1])
        isoform["starts"], isoform["ends"] = starts, ends
        for s, e in zip(starts, ends):
            assert s < e, (s, e)


def correct_boundaries_batch(side, jobs, majority_threshold, correction_window, ctx):
    """correct_boundaries() (:122-140) of every job: every read boundary within the window of an isoform boundary
    votes for its offset (GPU); an offset with at least the majority of the isoform's reads moves the boundary
    (the largest such offset wins, as the reference's ascending loop leaves it)."""
    if correction_window == 0:
        return
    assert side in ["starts", "ends"]
    isos, iso_read_off, iso_b_off, iso_bound, read_b_off, read_bound = [], [0], [0], [], [0], []
    for isoforms, _, reads in jobs:
        for isoform in isoforms.values():
            if side not in isoform:
                continue
            for rid in sorted(isoform["rids"]):
                read_bound.extend(reads[rid][side])              # KeyError when the split file lacks the read, as the reference
                read_b_off.append(len(read_bound))
            iso_bound.extend(isoform[side])
            isos.append(isoform)
            iso_read_off.append(len(read_b_off) - 1); iso_b_off.append(len(iso_bound))
    if not isos:
        return
    votes = ctx.boundary_votes(iso_read_off, iso_b_off, iso_bound, read_b_off, read_bound, correction_window)
    for i, isoform in enumerate(isos):
        n = len(isoform["rids"])
        for idx in range(iso_b_off[i + 1] - iso_b_off[i]):
            iso_s = isoform[side][idx]
            row = votes[iso_b_off[i] + idx].tolist()
            for k, v in enumerate(row):
                if v / n >= majority_threshold:
                    isoform[side][idx] = (k - correction_window) + iso_s


def get_gtf_records(isoforms):
    """[((chrom, first start), text)]: one transcript line and its exon lines per isoform that has exons (:72-119)."""
    records = []
    for (chrom, tint, pid, iid), isoform in isoforms.items():
        if "starts" not in isoform:
            continue
        starts, ends, strand = isoform["starts"], isoform["ends"], isoform["strand"]
        name = "{}_{}_{}".format(chrom, tint, iid)
        lines = ["\t".join([chrom, "freddie", "transcript", str(starts[0] + 1), str(ends[-1]), ".", strand, ".",
                            'transcript_id "{}"; read_support "{}";'.format(name, len(isoform["rids"]))])]
        for eid, (s, e) in enumerate(zip(starts, ends), start=1):
            lines.append("\t".join([chrom, "freddie", "exon", str(s), str(e), ".", strand, ".",
                                    'transcript_id "{0}"; exon_number "{1}"; exon_id "{0}_{1}"; '.format(name, eid)]))
        records.append(((chrom, starts[0]), "\n".join(lines)))
    return records


def run_consensus_batch(batch_args, ctx, verbose=True):
    """run_consensus() (:50-69) of several (contig, tint_id, cluster_tsv, split_tsv, majority_threshold,
    correction_window) tuples that share the last two values; returns the concatenated GTF records."""
    jobs = []
    for contig, tint_id, cluster_tsv, split_tsv, _, _ in batch_args:
        if verbose:
            print("Building isoforms for contig {}".format(contig))
        segments, reads, isoforms = read_cluster(cluster_tsv)
        jobs.append([isoforms, segments, reads, split_tsv])
    isoforms_cons_batch([(j[0], j[1], j[2]) for j in jobs], ctx)
    for j in jobs:
        read_split(j[3], j[2])
    if batch_args:
        m, w = batch_args[0][4], batch_args[0][5]
        assert all(a[4] == m and a[5] == w for a in batch_args)
        for side in ("starts", "ends"):
            correct_boundaries_batch(side, [(j[0], j[1], j[2]) for j in jobs], m, w, ctx)
    out = []
    for j in jobs:
        out.extend(get_gtf_records(j[0]))
    return out


def run_consensus(consensus_args, ctx=None):
    own = ctx is None
    ctx = ctx or Context(0)
    try:
        return run_consensus_batch([consensus_args], ctx)
    finally:
        if own:
            ctx.close()


def main(argv=None):
    args = parse_args(argv)
    consensus_args = []
    for contig in os.listdir(args.cluster_dir):
        if not os.path.isdir("{}/{}".format(args.cluster_dir, contig)):
            continue
        for cluster_tsv in glob.iglob("{}/{}/cluster_*.tsv".format(args.cluster_dir, contig)):
            tint_id = int(cluster_tsv[:-4].split("/")[-1].split("_")[-1])
            split_tsv = "{}/{}/split_{}_{}.tsv".format(args.split_dir, contig, contig, tint_id)
            assert os.path.isfile(split_tsv), split_tsv
            consensus_args.append([contig, tint_id, cluster_tsv, split_tsv, args.majority_threshold, args.correction_window])
    ctx = Context(args.device)
    gtf_records = []
    try:
        for i in range(0, len(consensus_args), max(1, args.batch_tints)):
            gtf_records.extend(run_consensus_batch(consensus_args[i:i + args.batch_tints], ctx))
    finally:
        ctx.close()
    gtf_records.sort()
    with open(args.output, "w+") as outfile:
        for _, record in gtf_records:
            outfile.write(record)
            outfile.write("\n")


if __name__ == "__main__":
    main()
