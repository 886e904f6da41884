"""Deterministic synthetic split partitions (inputs of the segmentation stage).

ctypes front-end of ``synth.c``.  The recipe follows SURVEY.md section 8(d); the file
formats are the split stage's (reference ``py/freddie_split.py:445-481``, ``:395-401``).
Used by the tests, by ``bench.py`` and by ``tests/golden/make_golden.py``; it is not part
of the segmentation path itself.
"""
import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfreddie_synth.so")
_SEED0 = 0xF4EDD1E


class _Params(ctypes.Structure):
    _fields_ = [
        ("seed", ctypes.c_uint64),
        ("n_reads", ctypes.c_int32),
        ("n_exons", ctypes.c_int32),
        ("n_isoforms", ctypes.c_int32),
        ("max_span", ctypes.c_int32),
        ("origin", ctypes.c_int32),
        ("tint_id", ctypes.c_int32),
        ("rid_base", ctypes.c_int32),
        ("indel_permille", ctypes.c_int32),
        ("keep_p", ctypes.c_double),
        ("rp", ctypes.c_double),
        ("jp", ctypes.c_double),
        ("jsd", ctypes.c_double),
    ]


def build(force=False):
    from .. import build as _build
    src = os.path.join(_HERE, "synth.c")
    _build.build_stamped(_SO, ["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, src], [src], force)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.fsynth_generate.restype = ctypes.c_void_p
        L.fsynth_generate.argtypes = [ctypes.POINTER(_Params), ctypes.c_int32]
        L.fsynth_free.argtypes = [ctypes.c_void_p]
        for name, rt in (("fsynth_n_reads", ctypes.c_int32), ("fsynth_n_intervals", ctypes.c_int32),
                         ("fsynth_n_exons", ctypes.c_int64), ("fsynth_n_cigar", ctypes.c_int64),
                         ("fsynth_seq_bytes", ctypes.c_int64)):
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [ctypes.c_void_p]
        vp = ctypes.c_void_p
        L.fsynth_copy_intervals.argtypes = [vp, vp, vp]
        L.fsynth_copy_exons.argtypes = [vp, vp, vp, vp, vp, vp]
        L.fsynth_copy_cigar.argtypes = [vp, vp, vp, vp]
        L.fsynth_copy_reads.argtypes = [vp, vp, vp, vp]
        L.fsynth_write_tsv.restype = ctypes.c_int
        L.fsynth_write_tsv.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
        _lib = L
    return _lib


@dataclass
class Partition:
    """One synthetic tint as flat arrays (genomic coordinates as the split stage writes them)."""
    contig: str
    tint_id: int
    rid_base: int
    iv_start: np.ndarray      # int32[K]   interval start
    iv_end: np.ndarray        # int32[K]   interval end (split convention; the stage treats it as a position)
    read_exon_off: np.ndarray  # int64[R+1]
    ex_ts: np.ndarray         # int32[I]
    ex_te: np.ndarray
    ex_qs: np.ndarray
    ex_qe: np.ndarray
    ex_cig_off: np.ndarray    # int64[I+1]
    cig_len: np.ndarray       # int32
    cig_op: np.ndarray        # uint8 (ASCII)
    strand: np.ndarray        # uint8 (ASCII '+'/'-')
    seq_off: np.ndarray       # int64[R+1]
    seq: np.ndarray           # uint8 (ASCII), empty when generated without sequences

    @property
    def n_reads(self):
        return len(self.strand)

    def read_name(self, i):
        return "synth_%d_%d" % (self.tint_id, i)


def partition_seed(index, seed0=_SEED0):
    return seed0 ^ int(index)


def generate(index=0, n_reads=200, n_exons=150, rp=0.05, jp=0.3, jsd=2.0, max_span=14, n_isoforms=8,
             keep_p=0.7, indel_permille=50, origin=100000, contig="chrS", with_seq=True, seed0=_SEED0,
             rid_base=None, write_dir=None):
    """Generate partition ``index``.  If ``write_dir`` is given, also write
    ``<write_dir>/<contig>/split_<contig>_<index>.tsv`` and ``reads_<contig>_<index>.tsv``."""
    L = lib()
    rb = index * 10_000_000 % 2_000_000_000 if rid_base is None else rid_base
    p = _Params(seed=partition_seed(index, seed0), n_reads=n_reads, n_exons=n_exons, n_isoforms=n_isoforms,
                max_span=max_span, origin=origin, tint_id=index, rid_base=rb, indel_permille=indel_permille,
                keep_p=keep_p, rp=rp, jp=jp, jsd=jsd)
    h = L.fsynth_generate(ctypes.byref(p), 1 if with_seq else 0)
    try:
        R = L.fsynth_n_reads(h)
        K = L.fsynth_n_intervals(h)
        I = L.fsynth_n_exons(h)
        NC = L.fsynth_n_cigar(h)
        SB = L.fsynth_seq_bytes(h)
        iv_s = np.empty(K, np.int32); iv_e = np.empty(K, np.int32)
        L.fsynth_copy_intervals(h, iv_s.ctypes.data, iv_e.ctypes.data)
        off = np.empty(R + 1, np.int64)
        ts = np.empty(I, np.int32); te = np.empty(I, np.int32); qs = np.empty(I, np.int32); qe = np.empty(I, np.int32)
        L.fsynth_copy_exons(h, off.ctypes.data, ts.ctypes.data, te.ctypes.data, qs.ctypes.data, qe.ctypes.data)
        coff = np.empty(I + 1, np.int64); clen = np.empty(NC, np.int32); cop = np.empty(NC, np.uint8)
        L.fsynth_copy_cigar(h, coff.ctypes.data, clen.ctypes.data, cop.ctypes.data)
        strand = np.empty(R, np.uint8); soff = np.empty(R + 1, np.int64); seq = np.empty(SB, np.uint8)
        L.fsynth_copy_reads(h, strand.ctypes.data, soff.ctypes.data, seq.ctypes.data if SB else None)
        if write_dir is not None:
            d = os.path.join(write_dir, contig)
            os.makedirs(d, exist_ok=True)
            rc = L.fsynth_write_tsv(h, os.path.join(d, "split_%s_%d.tsv" % (contig, index)).encode(),
                                    os.path.join(d, "reads_%s_%d.tsv" % (contig, index)).encode() if with_seq else None,
                                    contig.encode())
            if rc != 0:
                raise OSError("fsynth_write_tsv failed (%d)" % rc)
    finally:
        L.fsynth_free(h)
    return Partition(contig, index, rb, iv_s, iv_e, off, ts, te, qs, qe, coff, clen, cop, strand, soff, seq)


# Named workloads (SURVEY.md section 8d / BASELINE.json configs).  Values are keyword
# arguments of generate(); "n_partitions" partitions with indices base..base+n-1.
WORKLOADS = {
    # config 1, retention flavour: few long intervals, DP problems up to n~50
    "config1": dict(n_partitions=1, n_reads=200, n_exons=150, rp=0.05, jp=0.3, jsd=2.0),
    # config 1, exon-dense flavour: many tiny intervals, tiny DP problems
    "config1_dense": dict(n_partitions=1, n_reads=200, n_exons=150, rp=0.0, jp=0.3, jsd=2.0),
    # config 2: one partition, 50k reads, ~2k candidates in (almost) one interval
    "config2": dict(n_partitions=1, n_reads=50000, n_exons=1000, rp=0.05, jp=0.3, jsd=2.0),
    # config 3: 500 partitions x 1000 reads
    "config3": dict(n_partitions=500, n_reads=1000, n_exons=150, rp=0.05, jp=0.3, jsd=2.0),
    # config 4 (whole node): 4000 partitions x 500 reads; one GPU's share is 500 partitions
    "config4": dict(n_partitions=4000, n_reads=500, n_exons=150, rp=0.05, jp=0.3, jsd=2.0),
    # config 5: ONT-like error model, run with sigma=3.0, threshold_rate=0.80
    "config5": dict(n_partitions=5000, n_reads=1000, n_exons=150, rp=0.08, jp=0.8, jsd=6.0),
}
