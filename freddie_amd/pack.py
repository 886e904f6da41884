"""Flat (CSR) form of tint partitions: what fseg_upload() takes.

``read_reps`` follows the reference's grouping: reads with the same tuple of target exon
intervals form one rep, reps are kept in first-occurrence order (py/freddie_segment.py:165-170).
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class PackedPartition:
    iv_start: np.ndarray      # int32[K]
    iv_end: np.ndarray        # int32[K]
    rep_weight: np.ndarray    # int32[R]
    rep_exon_off: np.ndarray  # int64[R+1]
    ex_ts: np.ndarray         # int32[I]
    ex_te: np.ndarray         # int32[I]
    read_rep: np.ndarray      # int32[n_reads] rep index of each read (file order)

    @property
    def n_reps(self):
        return len(self.rep_weight)

    @property
    def n_reads(self):
        return len(self.read_rep)


def dedupe_reads(read_exon_off, ex_ts, ex_te):
    """Group reads by their exon tuple.  Returns (rep_weight, rep_exon_off, rep_ts, rep_te, read_rep)."""
    read_exon_off = np.asarray(read_exon_off, np.int64)
    ts = np.asarray(ex_ts, np.int32)
    te = np.asarray(ex_te, np.int32)
    n = len(read_exon_off) - 1
    key_to_rep = {}
    read_rep = np.empty(n, np.int32)
    first_read = []
    weights = []
    inter = np.empty(2 * len(ts), np.int32)
    inter[0::2] = ts
    inter[1::2] = te
    raw = inter.tobytes()
    for i in range(n):
        k = raw[8 * read_exon_off[i]:8 * read_exon_off[i + 1]]
        r = key_to_rep.get(k)
        if r is None:
            r = len(first_read)
            key_to_rep[k] = r
            first_read.append(i)
            weights.append(0)
        weights[r] += 1
        read_rep[i] = r
    first_read = np.asarray(first_read, np.int64)
    lens = (read_exon_off[first_read + 1] - read_exon_off[first_read]) if len(first_read) else np.empty(0, np.int64)
    rep_off = np.zeros(len(first_read) + 1, np.int64)
    np.cumsum(lens, out=rep_off[1:])
    idx = np.concatenate([np.arange(read_exon_off[f], read_exon_off[f + 1]) for f in first_read]) if len(first_read) \
        else np.empty(0, np.int64)
    return (np.asarray(weights, np.int32), rep_off, ts[idx].copy(), te[idx].copy(), read_rep)


def pack_partition(iv_start, iv_end, read_exon_off, ex_ts, ex_te, dedupe=True):
    if dedupe:
        w, off, ts, te, rr = dedupe_reads(read_exon_off, ex_ts, ex_te)
    else:
        n = len(read_exon_off) - 1
        w = np.ones(n, np.int32)
        off = np.asarray(read_exon_off, np.int64)
        ts = np.asarray(ex_ts, np.int32)
        te = np.asarray(ex_te, np.int32)
        rr = np.arange(n, dtype=np.int32)
    return PackedPartition(np.asarray(iv_start, np.int32), np.asarray(iv_end, np.int32), w, off, ts, te, rr)


def concat_batch(parts):
    """Concatenate PackedPartitions into the arrays of fseg_batch (include/freddie_seg.h)."""
    part_iv_off = np.zeros(len(parts) + 1, np.int64)
    part_rep_off = np.zeros(len(parts) + 1, np.int64)
    np.cumsum([len(p.iv_start) for p in parts], out=part_iv_off[1:])
    np.cumsum([p.n_reps for p in parts], out=part_rep_off[1:])
    iv_start = np.concatenate([p.iv_start for p in parts])
    iv_end = np.concatenate([p.iv_end for p in parts])
    rep_weight = np.concatenate([p.rep_weight for p in parts])
    ex_ts = np.concatenate([p.ex_ts for p in parts])
    ex_te = np.concatenate([p.ex_te for p in parts])
    rep_exon_off = np.zeros(part_rep_off[-1] + 1, np.int64)
    pos = 0
    base = 0
    for p in parts:
        n = p.n_reps
        rep_exon_off[pos:pos + n + 1] = p.rep_exon_off + base
        pos += n
        base += p.rep_exon_off[-1]
    return dict(part_iv_off=part_iv_off, iv_start=iv_start, iv_end=iv_end, part_rep_off=part_rep_off,
                rep_weight=rep_weight, rep_exon_off=rep_exon_off, ex_ts=ex_ts, ex_te=ex_te)
