"""Host side of the segmentation stage: the same interface as the reference's ``py/freddie_segment.py``
(CLI flags, ``split_*.tsv`` / ``reads_*.tsv`` in, ``segment_*.tsv`` out, and the in-process
``segment(tint, ...)`` seam), with all of the numeric work done by the gfx950 library behind the C-ABI
(``include/freddie_seg.h``).  There is no CPU implementation of the numeric path in this package.

Reference map (file:line of vpc-ccg/freddie ``py/freddie_segment.py``):
  parse_args :53-110 / main :847-885      -> parse_args(), main()
  read_split :121-171, read_sequence :174-185 -> read_split(), read_sequence()
  segment :738-844                        -> segment(), segment_batch()  (GPU)
  get_unaligned_gaps_and_polyA :370-472   -> unaligned_gaps_and_polyA()  (host; per-read string work)
  run_segment :681-735                    -> run_segment(), write_segment_tsv()
"""
import argparse
import glob
import multiprocessing
import os
import sys
from math import ceil

import numpy as np

from . import pack, tables

CIGAR_OPS = "MIDNSHPX="


# --------------------------------------------------------------------------------------------------
# arguments (same flags, defaults and range checks as the reference; --gpus / --batch-reads are additions)
# --------------------------------------------------------------------------------------------------
MAX_PROBLEM_SIZE = 1000      # -mps: the library's largest DP problem is 1 024 candidates (kNGiant, freddie_amd/csrc/freddie_seg.hip)


def str_to_bool(value):
    if isinstance(value, bool):
        return value
    v = value.lower()
    if v in ("false", "f", "0", "no", "n"):
        return False
    if v in ("true", "t", "1", "yes", "y"):
        return True
    raise ValueError("{} is not a valid boolean value".format(value))


def parse_args(argv=None):
    # (allow_abbrev=False: py/freddie_segment.py reads --gpus / --devices off the command line before this parser runs, to
    # start the one GPU's contexts early; an abbreviated "--dev 0,1" would be accepted here and missed there)
    ap = argparse.ArgumentParser(description="Cluster aligned reads into isoforms", allow_abbrev=False)
    ap.add_argument("-s", "--split-dir", type=str, required=True, help="Path to Freddie split directory of the reads")
    ap.add_argument("--consider-ends", type=str_to_bool, nargs="?", const=True, default=False,
                    help="Consider the start and end splice sites in segmentation")
    ap.add_argument("-o", "--outdir", type=str, default="freddie_segment/",
                    help="Path to output directory. Default: freddie_segment/")
    ap.add_argument("-t", "--threads", type=int, default=1,
                    help="Host threads per GPU for parsing / annotating / writing. Default: 1")
    ap.add_argument("-sd", "--sigma", type=float, default=5.0, help="Sigma value for gaussian_filter1d")
    ap.add_argument("-tp", "--threshold-rate", type=float, default=0.90,
                    help="Threshold rate above which the read will be considered as covering a segment. Default: 0.9")
    ap.add_argument("-vf", "--variance-factor", type=float, default=3.0,
                    help="The stdev factor to fix a candidate peak. Default 3.0")
    ap.add_argument("-mps", "--max-problem-size", type=int, default=50,
                    help="Maximum number of candidate breakpoints allowed per segmentation problem")
    ap.add_argument("-lo", "--min-read-support-outside", type=int, default=3,
                    help="Minimum reads support for splice site to support a breakpoint")
    ap.add_argument("--gpus", type=int, default=0, help="Number of GPUs to scatter partitions over (0 = all visible)")
    ap.add_argument("--devices", type=str, default=None,
                    help="Comma-separated GPU ordinals, one worker per entry (overrides --gpus; e.g. 0,1,2,3)")
    ap.add_argument("--batch-reads", type=int, default=250000, help="Reads per device batch")
    ap.add_argument("--sidecar", choices=("auto", "off", "write"), default="auto",
                    help="Binary side-cars (split_*.fsc) of the split TSVs: auto = use the fresh ones that exist; "
                         "write = also emit them for partitions that had to be parsed; off = always parse the TSVs")
    args = ap.parse_args(argv)
    assert 1 >= args.threshold_rate >= 0.5
    assert 10 > args.variance_factor > 0
    assert 50 >= args.sigma > 0
    assert args.max_problem_size > 3
    # (the reference has no upper bound -- its optimize() is O(n^3 R) Python --; this library's DP kernels take problems of up to
    # 1 024 candidates, and break_large_problems leaves problems a little larger than the limit it is given)
    if args.max_problem_size > MAX_PROBLEM_SIZE:
        ap.error("--max-problem-size %d is beyond what this implementation runs (at most %d)" % (args.max_problem_size, MAX_PROBLEM_SIZE))
    assert args.min_read_support_outside >= 0
    assert args.threads > 0
    return args


# --------------------------------------------------------------------------------------------------
# input files
# --------------------------------------------------------------------------------------------------
def _parse_cigar(text):
    ops = []
    num = 0
    have = False
    for ch in text:
        if "0" <= ch <= "9":
            num = num * 10 + (ord(ch) - 48)
            have = True
        else:
            if not have or ch not in CIGAR_OPS:
                raise ValueError("bad CIGAR %r" % text)
            ops.append((num, ch))
            num = 0
            have = False
    if have or not ops:
        raise ValueError("bad CIGAR %r" % text)
    return ops


def _parse_read_interval(field):
    try:
        t, q, cig = field.split(":")
        ts, te = t.split("-")
        qs, qe = q.split("-")
        return int(ts), int(te), int(qs), int(qe), _parse_cigar(cig)
    except ValueError as exc:
        raise ValueError("bad read interval field %r" % field) from exc


def read_split(split_tsv):
    """One ``split_<contig>_<tint>.tsv`` -> list of tint dicts (the reference asserts there is exactly one)."""
    tints = {}
    with open(split_tsv) as f:
        for line in f:
            if not line.endswith("\n"):
                raise ValueError("%s: line without newline" % split_tsv)
            cols = line[:-1].split("\t")
            if line[0] == "#":
                if len(cols) != 4:
                    raise ValueError("%s: bad header line" % split_tsv)
                intervals = []
                for piece in cols[2].split(","):
                    s, e = piece.split("-")
                    intervals.append((int(s), int(e)))
                tint = dict(id=int(cols[1]), chr=cols[0][1:], intervals=intervals, read_count=int(cols[3]),
                            reads=[], read_reps={})
                assert tint["id"] not in tints, "Transcriptional interval with id {} is repeated!".format(tint["id"])
                assert all(a[1] < b[0] for a, b in zip(intervals[:-1], intervals[1:])), intervals
                assert all(s < e for s, e in intervals)
                tints[tint["id"]] = tint
            else:
                if len(cols) < 6 or cols[3] not in ("+", "-"):
                    raise ValueError("%s: bad read line" % split_tsv)
                read = dict(id=int(cols[0]), name=cols[1], chr=cols[2], strand=cols[3], tint=int(cols[4]),
                            intervals=[_parse_read_interval(x) for x in cols[5:]])
                iv = read["intervals"]
                assert all(a[1] <= b[0] and a[3] <= b[2] for a, b in zip(iv[:-1], iv[1:]))
                assert all(x[0] < x[1] and x[2] < x[3] for x in iv)
                tints[read["tint"]]["reads"].append(read)
    for tint in tints.values():
        assert len(tint["reads"]) == tint["read_count"]
        reps = {}
        for ridx, read in enumerate(tint["reads"]):
            reps.setdefault(tuple((x[0], x[1]) for x in read["intervals"]), []).append(ridx)
        tint["read_reps"] = list(reps.items())
    return list(tints.values())


def read_sequence(tint, reads_tsv):
    seqs = {}
    with open(reads_tsv) as f:
        for line in f:
            cols = line.rstrip().split("\t")
            seqs[int(cols[0])] = cols[3]
    assert len(seqs) == len(tint["reads"]), tint["id"]
    for read in tint["reads"]:
        read["seq"] = seqs[read["id"]]
        read["length"] = len(read["seq"])


# --------------------------------------------------------------------------------------------------
# per-read string work (host): soft clips, poly-A/T tails and unaligned gaps between label-1 runs
# --------------------------------------------------------------------------------------------------
_COMPLEMENT = {"A": "T", "C": "G", "G": "C", "T": "A"}


def _thread_cigar(cigar, t_goal, t_pos, q_pos):
    """Query position reached when the target position advances from t_pos to t_goal (:289-304).
    The step of every op, insertions included, is clipped to the remaining target distance."""
    assert t_pos <= t_goal
    idx = 0
    while t_pos < t_goal:
        length, op = cigar[idx]
        step = min(length, t_goal - t_pos)
        if op in "MX=":
            t_pos += step
            q_pos += step
        elif op == "D":
            t_pos += step
        elif op == "I":
            q_pos += step
        idx += 1
    assert t_pos == t_goal
    return q_pos


def _query_at_or_after(start, read):
    """First query position aligned at or after target ``start`` (+ non-positive slack) (:307-326)."""
    for t_start, t_end, q_start, q_end, cigar in read["intervals"]:
        if t_end < start:
            continue
        if start < t_start:
            q_pos, slack = q_start, start - t_start
        else:
            q_pos, slack = _thread_cigar(cigar, start, t_start, q_start), 0
        assert slack <= 0 and q_start <= q_pos <= q_end
        return q_pos, slack
    raise AssertionError("no exon at or after %d" % start)


def _query_at_or_before(end, read):
    """Last query position aligned at or before target ``end`` (+ non-positive slack) (:329-349)."""
    for t_start, t_end, q_start, q_end, cigar in reversed(read["intervals"]):
        if t_start > end:
            continue
        if t_end < end:
            q_pos, slack = q_end, t_end - end
        else:
            q_pos, slack = _thread_cigar(cigar, end, t_start, q_start), 0
        assert slack <= 0 and 0 <= q_pos <= q_end
        return q_pos, slack
    raise AssertionError("no exon at or before %d" % end)


def _poly_runs(seq, s, e, step, char):
    """Local-alignment runs (+1 match, -2 mismatch, floor 0) of ``char`` over seq[s:e:step] (:352-367).
    Yields (first index, length up to the best score, purity)."""
    if e - s == 0:
        return
    window = seq[s:e:step]
    score = 1 if seq[s] == char else 0
    scores = [score]
    for c in seq[s + step:e:step]:
        score = max(0, score + (1 if c == char else -2))
        scores.append(score)
    i, n = 0, len(scores)
    while i < n:
        if scores[i] <= 0:
            i += 1
            continue
        j = i
        best_s, best_i = scores[i], i
        while j < n and scores[j] > 0:
            if scores[j] >= best_s:              # max over (score, index): later index wins ties
                best_s, best_i = scores[j], j
            j += 1
        length = best_i + 1 - i
        yield i, length, window[i:i + length].count(char) / length
        i = j


def unaligned_gaps_and_polyA(read, segs):
    """Sets read['gaps'] from read['data'] (labels), the read's alignment and sequence (:370-472)."""
    read["gaps"] = set()
    data = read["data"]
    if 1 not in data:
        return
    runs = []
    i, n = 0, len(data)
    while i < n:
        if data[i] != 1:
            i += 1
            continue
        j = i
        while j + 1 < n and data[j + 1] == 1:
            j += 1
        runs.append((i, j))
        i = j + 1
    q_ssc, _ = _query_at_or_after(segs[runs[0][0]][0], read)
    q_esc, _ = _query_at_or_before(segs[runs[-1][1]][1], read)
    length = read["length"]
    assert 0 <= q_ssc <= q_esc <= length
    minus = read["strand"] == "-"
    gaps = read["gaps"]

    def scan(s, e):
        found = []
        for char in ("A", "T"):
            if minus:
                runs_ = _poly_runs(read["seq"], -s - 1, -e - 1, -1, _COMPLEMENT[char])
            else:
                runs_ = _poly_runs(read["seq"], s, e, 1, char)
            for first, ln, purity in runs_:
                if ln >= 20 and purity >= 0.85:
                    found.append((first, ln, purity, char))
        best = None
        for cand in found:                       # max purity, first one wins ties
            if best is None or cand[2] > best[2]:
                best = cand
        return best

    best = scan(0, q_ssc)
    if best is not None:
        first, ln, _, char = best
        assert 0 <= first < q_ssc
        gap = q_ssc - first - ln
        assert 0 <= gap < q_ssc
        gaps.add("S{}_{}:{}".format(char, ln, gap))
        gaps.add("SSC:{}".format(first))
    else:
        gaps.add("SSC:{}".format(q_ssc))
    best = scan(q_esc, length)
    if best is not None:
        first, ln, _, char = best
        assert 0 <= first < length - q_esc
        gaps.add("E{}_{}:{}".format(char, ln, first))
        gaps.add("ESC:{}".format(length - q_esc - first))
        assert length - q_esc - first > 0
    else:
        gaps.add("ESC:{}".format(length - q_esc))
    for (_, last1), (first2, _) in zip(runs[:-1], runs[1:]):
        q_a, slack_a = _query_at_or_before(segs[last1][1], read)
        q_b, slack_b = _query_at_or_after(segs[first2][0], read)
        assert 0 < q_a <= q_b < length
        size = max(0, q_b - q_a + slack_a + slack_b)
        assert 0 <= size < length and last1 < first2
        gaps.add("{}-{}:{}".format(last1, first2, size))
    read["gaps"] = sorted(gaps)


# --------------------------------------------------------------------------------------------------
# the numeric path (GPU)
# --------------------------------------------------------------------------------------------------
def pack_tint(tint):
    """tint dict -> flat arrays (pack.PackedPartition) in the reference's read_reps order."""
    iv = np.asarray(tint["intervals"], dtype=np.int32).reshape(-1, 2)
    reps = tint["read_reps"]
    off = np.zeros(len(reps) + 1, np.int64)
    np.cumsum([len(k) for k, _ in reps], out=off[1:])
    flat = np.asarray([x for k, _ in reps for x in k], dtype=np.int32).reshape(-1, 2)
    read_rep = np.empty(len(tint["reads"]), np.int32)
    for ri, (_, ridxs) in enumerate(reps):
        read_rep[ridxs] = ri
    return pack.PackedPartition(np.ascontiguousarray(iv[:, 0]), np.ascontiguousarray(iv[:, 1]),
                                np.asarray([len(r) for _, r in reps], np.int32), off,
                                np.ascontiguousarray(flat[:, 0]), np.ascontiguousarray(flat[:, 1]), read_rep)


_default_ctx = {}


def default_context(device=0):
    from . import _lib
    if device not in _default_ctx:
        _default_ctx[device] = _lib.Context(device)
    return _default_ctx[device]


def segment_batch(tints, sigma, smoothed_threshold, threshold_rate, variance_factor, max_problem_size,
                  min_read_support_outside, ignore_ends, ctx=None, gaps=True):
    """segment() over a batch of tints in one device pass.  Mutates every tint like the reference's segment():
    tint['final_positions'], tint['segs'], read['data'], read['gaps']."""
    ctx = ctx or default_context()
    parts = [pack_tint(t) for t in tints]
    ctx.set_params(sigma, threshold_rate, variance_factor, max_problem_size, min_read_support_outside, ignore_ends,
                   tables.gaussian_half_kernel(sigma, 4.0), tables.gaussian_half_kernel(sigma, 1.0),
                   np.asarray(smoothed_threshold, np.float64))
    ctx.upload(**pack.concat_batch(parts))
    ctx.run()
    part_final_off, final_pos, label_off, labels = ctx.download()
    for p, (tint, part) in enumerate(zip(tints, parts)):
        fp = final_pos[part_final_off[p]:part_final_off[p + 1]]
        tint["final_positions"] = fp.tolist()
        tint["segs"] = list(zip(tint["final_positions"][:-1], tint["final_positions"][1:]))
        S = len(fp) - 1
        lab = (labels[label_off[p]:label_off[p + 1]].reshape(part.n_reps, S) - 48).astype(np.uint8)
        rows = [row.tolist() for row in lab]
        for ri, (_, ridxs) in enumerate(tint["read_reps"]):
            for ridx in ridxs:
                tint["reads"][ridx]["data"] = list(rows[ri])
        if gaps:
            for read in tint["reads"]:
                unaligned_gaps_and_polyA(read, tint["segs"])
    return [t["id"] for t in tints]


def segment(tint, sigma, smoothed_threshold, threshold_rate, variance_factor, max_problem_size,
            min_read_support_outside, ignore_ends, ctx=None):
    """Same signature and effects as the reference's segment() (:738-844)."""
    return segment_batch([tint], sigma, smoothed_threshold, threshold_rate, variance_factor, max_problem_size,
                         min_read_support_outside, ignore_ends, ctx=ctx)[0]


# --------------------------------------------------------------------------------------------------
# output + driver
# --------------------------------------------------------------------------------------------------
def write_segment_tsv(tint, path):
    with open(path, "w+") as out:
        out.write("#{}\t{}\t{}\n".format(tint["chr"], tint["id"], ",".join(map(str, tint["final_positions"]))))
        for read in tint["reads"]:
            out.write("\t".join((str(read["id"]), read["name"], read["chr"], read["strand"], str(read["tint"]),
                                 "".join(map(str, read["data"])), "".join(g + "," for g in read["gaps"]))))
            out.write("\n")


def _load_partition(split_dir, contig, tint_id):
    tints = read_split("{}/{}/split_{}_{}.tsv".format(split_dir, contig, contig, tint_id))
    assert len(tints) == 1
    read_sequence(tints[0], "{}/{}/reads_{}_{}.tsv".format(split_dir, contig, contig, tint_id))
    return tints[0]


def _job_paths(job):
    split_dir, outdir, contig, tint_id = job
    return ("{}/{}/split_{}_{}.tsv".format(split_dir, contig, contig, tint_id),
            "{}/{}/reads_{}_{}.tsv".format(split_dir, contig, contig, tint_id),
            "{}/{}/segment_{}_{}.tsv".format(outdir, contig, contig, tint_id),
            "{}/{}/segment_{}_{}.log".format(outdir, contig, contig, tint_id))


def set_context_params(ctx, params):
    (sigma, smoothed_threshold, threshold_rate, variance_factor, max_problem_size, min_read_support_outside,
     ignore_ends) = params
    ctx.set_params(sigma, threshold_rate, variance_factor, max_problem_size, min_read_support_outside, ignore_ends,
                   tables.gaussian_half_kernel(sigma, 4.0), tables.gaussian_half_kernel(sigma, 1.0),
                   np.asarray(smoothed_threshold, np.float64))


def _early_mod():
    from . import _early
    return _early


def sidecar_path(split_tsv):
    """split_<contig>_<tint>.tsv -> split_<contig>_<tint>.fsc (binary side-car, include/freddie_host.h)."""
    return split_tsv[:-4] + ".fsc"


def load_batch_native(jobs, threads=1, sidecar="off"):
    """Load the partitions of a batch with the native host library (multi-threaded): from their binary side-cars
    where those are fresh (``sidecar`` auto / write), else by parsing the TSVs."""
    from . import _host
    paths = [_job_paths(j) for j in jobs]
    _host.touch([p[3] for p in paths], n_threads=min(threads, 8))      # the reference leaves an empty .log per partition (:695)
    if sidecar == "off":
        return _host.HostBatch([p[0] for p in paths], [p[1] for p in paths], n_threads=threads)
    scs = [sidecar_path(p[0]) for p in paths]
    hb = _host.HostBatch([p[0] for p in paths], [p[1] for p in paths], n_threads=threads, sidecar_paths=scs)
    if sidecar == "write" and hb.n_from_sidecar < hb.n_part:
        hb.write_sidecars(scs, n_threads=threads)
    return hb


def run_segment_batch(jobs, params, ctx, threads=1, host_batch=None, params_set=False):
    """jobs: list of (split_dir, outdir, contig, tint_id).  Parses (natively), segments on the GPU, annotates and
    writes the outputs (natively).  Returns (host_batch, results) for the caller to write when ``defer_write``."""
    hb = host_batch or load_batch_native(jobs, threads)
    try:
        if not params_set:
            set_context_params(ctx, params)
        ctx.upload(**hb.arrays())
        ctx.run()
        part_final_off, final_pos, label_off, labels = ctx.download()
        hb.write(part_final_off, final_pos, label_off, labels, [_job_paths(j)[2] for j in jobs], n_threads=threads)
    finally:
        hb.close()
    return [(j[2], j[3]) for j in jobs]


def run_segment_batch_python(jobs, params, ctx):
    """The same through the Python host code (reference-shaped dicts); kept for the in-process seam and tests."""
    tints = []
    for split_dir, outdir, contig, tint_id in jobs:
        open("{}/{}/segment_{}_{}.log".format(outdir, contig, contig, tint_id), "w+").close()
        tints.append(_load_partition(split_dir, contig, tint_id))
    segment_batch(tints, *params, ctx=ctx)
    for (split_dir, outdir, contig, tint_id), tint in zip(jobs, tints):
        write_segment_tsv(tint, "{}/{}/segment_{}_{}.tsv".format(outdir, contig, contig, tint_id))
    return [(j[2], j[3]) for j in jobs]


def batch_too_large(exc):
    """True for the errors of Context.upload / run that a smaller batch cures (not for an input the reference rejects too)."""
    code = getattr(exc, "code", None)
    msg = str(exc).lower()
    return (code == 4 and "split it" in msg) or (code == 2 and "memory" in msg)


def run_batches(batches, params, ctxs, threads, on_done, sidecar="off"):
    """Pipelined driver of one GPU.  ``ctxs``: one or two contexts on the same device (a single Context is accepted).
    Batch i+1 (and i+2) are being parsed by the native loader while batch i is on the device and batch i-1 is annotated
    and written; with two contexts consecutive batches alternate between them, so one batch's upload and the other's
    download overlap the kernels of the batch in between (each context has its own stream, pinned staging and pinned
    results, and the writer reads the results in place).  FREDDIE_TIMING=1 prints one line per batch to stderr (load /
    device / write seconds, partitions taken from side-cars)."""
    import threading
    import time
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    if not batches:
        return
    timing = os.environ.get("FREDDIE_TIMING") == "1"
    done_lock = threading.Lock()

    t_pipe = time.perf_counter()
    marks = []                                   # FREDDIE_TIMING=1: (what, batch, start, end) relative to the pipeline's start

    def load(jobs, i=-1):
        t0 = time.perf_counter()
        hb = load_batch_native(jobs, threads, sidecar)
        t1 = time.perf_counter()
        if timing:
            marks.append(("load", i, t0 - t_pipe, t1 - t_pipe))      # (i = the batch's index; -1: a half of a batch that was too large)
        return hb, t1 - t0

    def write(hb, res, jobs, t_load, t_dev, i):
        t0 = time.perf_counter()
        try:
            hb.write(*res, [_job_paths(j)[2] for j in jobs], n_threads=threads, packed=True)
        finally:
            n_sc, n_reads = hb.n_from_sidecar, hb.n_reads
            hb.close()
        if timing:
            marks.append(("write", i, t0 - t_pipe, time.perf_counter() - t_pipe))
            print("[freddie_segment] batch %d: %d partitions (%d from side-cars), %d reads: load %.3f s, device %.3f s, "
                  "write %.3f s" % (i, len(jobs), n_sc, n_reads, t_load, t_dev, time.perf_counter() - t0), file=sys.stderr)
        with done_lock:
            on_done(len(jobs))               # (a count per batch, not a message per partition: see main())

    with ThreadPoolExecutor(max_workers=2) as load_pool, ThreadPoolExecutor(max_workers=2) as dev_pool, \
            ThreadPoolExecutor(max_workers=2) as write_pool:
        loads = deque()
        nxt = 0

        def prefetch():
            nonlocal nxt
            # (two being parsed, three waiting or done: while the contexts come up -- 0.08-0.15 s -- nobody takes a parsed batch
            # away, and with three in all the loaders sat idle for the second half of that time (FREDDIE_TIMING=1, round 6).  Host
            # memory: a parsed batch is its mapped TSVs plus ~100 B per read of arrays -- with --batch-reads 250 000 about 25 MB
            # each, five of them held here, two more on the contexts and up to two with the writers: ~0.25 GB at the default)
            while nxt < len(batches) and len(loads) < 5:
                loads.append(load_pool.submit(load, batches[nxt], nxt))
                nxt += 1

        prefetch()                           # the first batches are being parsed while the contexts come up
        if hasattr(ctxs, "result"):
            ctxs = ctxs.result()             # a future of open_contexts()
            if timing:
                marks.append(("contexts-up", 0, 0.0, time.perf_counter() - t_pipe))
        if not isinstance(ctxs, (list, tuple)):
            ctxs = [ctxs]
        for ctx in ctxs:
            set_context_params(ctx, params)
        n_ctx = len(ctxs)
        last_write = [None] * n_ctx          # the write that still reads context k's pinned results

        def device(k, i, jobs, hb, t_load):
            ctx = ctxs[k]
            t0 = time.perf_counter()
            try:
                ctx.upload(**hb.arrays())
                ctx.run()
                t1 = time.perf_counter()
                if last_write[k] is not None:
                    last_write[k].result()   # its results live in this context's pinned buffers until it has finished
                t2 = time.perf_counter()
                res = ctx.results(packed=True)    # two bits per label across PCIe; the writer unpacks rows into the TSV
            except BaseException as exc:
                hb.close()
                if len(jobs) > 1 and batch_too_large(exc):
                    # Batches are cut by split-file bytes; many low-coverage partitions (few bytes, thousands of positions
                    # each) can exceed what one upload takes (2^31 positions / lanes, or device memory).  The reference works
                    # partition by partition, so nothing but the batching is at fault: halve and go on.
                    for half in (jobs[:len(jobs) // 2], jobs[len(jobs) // 2:]):
                        hb2, t_load2 = load(half)
                        device(k, i, half, hb2, t_load2)
                    return
                raise
            t_dev = (t1 - t0) + (time.perf_counter() - t2)       # without the wait for the writer
            if timing:
                marks.append(("device", i, t0 - t_pipe, time.perf_counter() - t_pipe))
            last_write[k] = write_pool.submit(write, hb, res, jobs, t_load, t_dev, i)

        dev_futs = [None] * n_ctx
        try:
            for i, jobs in enumerate(batches):
                hb, t_load = loads.popleft().result()
                prefetch()
                k = i % n_ctx
                if dev_futs[k] is not None:
                    dev_futs[k].result()     # the context is free again (and its error, if any, surfaces here)
                dev_futs[k] = dev_pool.submit(device, k, i, jobs, hb, t_load)
            for f in dev_futs:
                if f is not None:
                    f.result()
            for f in last_write:
                if f is not None:
                    f.result()
            if timing:
                print("[freddie_segment] pipeline (s from its start): " + "; ".join(
                    "%s %d %.3f-%.3f" % m for m in sorted(marks, key=lambda m: m[2])), file=sys.stderr)
        finally:
            for f in loads:                  # an error above: do not leak the batches that were already parsed
                try:
                    f.result()[0].close()
                except Exception:
                    pass


def run_segment(segment_args, ctx=None):
    """Reference-shaped entry: one 11-tuple as built by the reference's main() (:858-870)."""
    (split_dir, outdir, contig, tint_id, sigma, smoothed_threshold, threshold_rate, variance_factor,
     max_problem_size, min_read_support_outside, ignore_ends) = segment_args
    params = (sigma, smoothed_threshold, threshold_rate, variance_factor, max_problem_size,
              min_read_support_outside, ignore_ends)
    return run_segment_batch([(split_dir, outdir, contig, tint_id)], params, ctx or default_context())[0]


def discover(split_dir, outdir):
    """(contig, tint_id, cost) of every partition, creating the output directories (:852-857)."""
    # (the listing and the sizes natively -- readdir per contig directory, the stat calls on eight threads, include/freddie_host.h:
    # as a Python loop over scandir + stat the 8 000 files of the 2 M-read job were 0.04-0.14 s of its 0.8 s on the GPU box)
    from . import _host
    if not os.path.isdir(split_dir):
        raise FileNotFoundError(split_dir)
    contigs, found = _host.discover(split_dir)
    for contig in contigs:
        os.makedirs("{}/{}".format(outdir, contig), exist_ok=True)
    return found


def make_batches(jobs_with_cost, bytes_per_batch):
    """Consecutive jobs up to bytes_per_batch of split TSV each -- but at least eight batches once there is enough input,
    so that loading, the device and writing overlap and the pipeline's fill and drain (the first batch's load, the last
    one's write) stay short; a batch costs the device a few milliseconds, so small batches are cheap."""
    total = sum(c for _, c in jobs_with_cost)
    bytes_per_batch = min(bytes_per_batch, max(total // 8, 8 << 20))
    # ... and the first batches are cut smaller (a quarter, a quarter, a half, a half of the size): the device and the writers have
    # nothing to do until the first load is through, and the first load is what the pipeline's start costs (0.10-0.13 s of the
    # 0.55 s the 2 M-read job spends in its batches, FREDDIE_TIMING=1)
    ramp = [4, 4, 2, 2] if total > 4 * bytes_per_batch else []
    batches, cur, size = [], [], 0
    for job, cost in jobs_with_cost:
        limit = bytes_per_batch // ramp[len(batches)] if len(batches) < len(ramp) else bytes_per_batch
        if cur and size + cost > limit:
            batches.append(cur)
            cur, size = [], 0
        cur.append(job)
        size += cost
    if cur:
        batches.append(cur)
    return batches


# One worker process per GPU.  "spawn": a worker must not inherit anything of the parent's (the parent never touches
# the GPU, see devices.py, but a library that did would make fork unsafe).  The CPU test of main() switches to "fork" so
# that its stand-in for open_contexts() reaches the workers.
WORKER_START_METHOD = "spawn"


def open_contexts(device, n=2):
    """The contexts one GPU worker drives: two per device, so that copies and kernels of consecutive batches overlap."""
    from . import _early, _lib
    early = _early.take(device)              # what the drop-in script started before the imports, if it did
    ctxs = []
    try:
        while early and len(ctxs) < n:
            ctxs.append(_lib.Context(device, handle=early[0]))      # (a stale library is refused here: _lib.load())
            early.pop(0)
        while len(ctxs) < n:
            ctxs.append(_lib.Context(device))
    except BaseException:
        for c in ctxs:
            c.close()
        _early.discard(early)
        raise
    _early.discard(early)
    return ctxs


def _gpu_worker(device, n_workers, jobs_with_cost, params, batch_bytes, threads, queue, sidecar="off", worker=None):
    """jobs_with_cost: the worker's jobs, or the receiving end of a pipe they arrive through (main() starts its workers before
    it has looked at the split directory: interpreter start-up, imports and the GPU's contexts -- 0.3-0.5 s -- run beside the
    parent's discovery and scatter instead of behind them)."""
    from . import devices
    from concurrent.futures import ThreadPoolExecutor
    # host threads of this worker stay on its share of the cores (next to its GPU where that is known: devices.py)
    devices.pin_worker(device if worker is None else worker, n_workers, device)
    ok = False
    with ThreadPoolExecutor(max_workers=1) as boot:
        ctx_future = boot.submit(open_contexts, device)
        try:
            if hasattr(jobs_with_cost, "recv"):
                conn = jobs_with_cost
                try:
                    jobs_with_cost = _expand_jobs(conn.recv())
                finally:
                    conn.close()
            run_batches(make_batches(jobs_with_cost, batch_bytes), params, ctx_future, threads, queue.put, sidecar)
            ok = True
        finally:
            try:
                for ctx in ctx_future.result():
                    ctx.close()
            finally:
                queue.put(None)
    if ok and WORKER_START_METHOD == "spawn" and _early_mod().fast_exit_allowed():
        # (a spawned worker that has finished its share leaves without the interpreter's and the HIP runtime's tear-down, like the
        # drop-in script: py/freddie_segment.py -- once its last messages are on their way to the parent)
        queue.close()
        queue.join_thread()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def _expand_jobs(msg):
    """(split_dir, outdir, [(contig, tint_id, cost), ...]) -> [((split_dir, outdir, contig, tint_id), cost), ...]; None: no work
    (the parent stopped before it had a job list: an error it reports itself)."""
    if msg is None:
        return []
    split_dir, outdir, items = msg
    return [((split_dir, outdir, contig, tint_id), cost) for contig, tint_id, cost in items]


def main(argv=None, leave_contexts=False):
    """leave_contexts: the caller ends the process right after main() (the drop-in script's fast exit): a successful one-GPU run then
    leaves its contexts to the process's end instead of closing them one by one (streams, events, slabs: 0.03-0.06 s)."""
    import time
    t_start = time.perf_counter()
    from . import devices, scatter
    args = parse_args(argv)
    split_dir = args.split_dir.rstrip("/")
    if args.devices:
        device_list = [int(x) for x in args.devices.split(",") if x != ""]
    else:
        n_gpus = args.gpus
        if n_gpus <= 0:
            # counted from sysfs / the runtime's environment variables, NOT through the HIP runtime: this process may
            # have to start one worker process per GPU, which a process that has initialised HIP must not do
            n_gpus = devices.visible_gpu_count()
        device_list = list(range(n_gpus))
    n_gpus = len(device_list)
    if n_gpus <= 0:
        raise SystemExit("freddie_segment: no GPU visible (this implementation has no CPU path)")
    from . import _early
    if n_gpus != 1 and _early.started():
        # the CLI shim took the command line for a one-GPU run and this process has begun HIP's start-up: it must not start
        # worker processes now (a spawn is an exec, and a process that has initialised the GPU must not exec: devices.py)
        raise SystemExit("freddie_segment: --gpus / --devices name %d GPUs but were first read as one (repeated or "
                         "contradictory options?): give each once, in full" % n_gpus)
    # One GPU: this process drives it itself, and bringing its contexts up (library load, HIP start-up, streams: a few tenths
    # of a second, a third of the wall time of a 2 M-read job) starts NOW, beside the directory scan and the first parse.
    boot = ctx_future = None
    if n_gpus == 1:
        from concurrent.futures import ThreadPoolExecutor
        boot = ThreadPoolExecutor(max_workers=1)
        ctx_future = boot.submit(open_contexts, device_list[0])
    def close_boot():
        # whatever stops main() before the batches run (a missing split directory, say): the contexts that are coming up
        # beside it are closed and the executor ended, so the user sees the error at once and not after HIP's start-up
        if boot is not None:
            boot.shutdown(wait=True)
            if ctx_future is not None and ctx_future.exception() is None:
                for ctx in ctx_future.result():
                    ctx.close()
    params = (args.sigma, tables.smooth_threshold(args.threshold_rate), args.threshold_rate, args.variance_factor,
              args.max_problem_size, args.min_read_support_outside, not args.consider_ends)
    batch_bytes = max(1, args.batch_reads) * 1400          # ~1.4 KB of split TSV per read
    # Several GPUs: one worker process each, started NOW -- before this process has looked at the split directory -- with the
    # sending end of a pipe kept here: a worker's interpreter, imports and GPU contexts (0.3-0.5 s) come up beside the
    # discovery and the scatter, and its job list follows through the pipe.  (Until round 5 the workers were started one after
    # the other once the scatter was known, each with its job list pickled into the start-up message.)
    procs, conns, queue = [], [], None
    if n_gpus > 1:
        mp = multiprocessing.get_context(WORKER_START_METHOD)
        queue = mp.Queue()
        for w, dev in enumerate(device_list):
            recv_end, send_end = mp.Pipe(duplex=False)
            pr = mp.Process(target=_gpu_worker, args=(dev, n_gpus, recv_end, params, batch_bytes, args.threads, queue, args.sidecar, w))
            pr.start()
            recv_end.close()
            procs.append(pr)
            conns.append(send_end)
    t_started = time.perf_counter()
    try:
        parts = discover(split_dir, args.outdir)
        costs = [c for _, _, c in parts]
        assign = scatter.lpt_scatter(costs, n_gpus)
    except BaseException:
        close_boot()
        for conn in conns:                       # the workers end at once (nothing to do) and the error is this process's to report
            try:
                conn.send(None); conn.close()
            except OSError:
                pass
        for pr in procs:
            pr.join()
        raise
    total = len(parts)
    step = ceil(total / 100) if total else 1
    done_count = 0

    def report(n=1):
        # (the reference prints as its pool hands partitions back, one line per hundredth of them, :877; here partitions come
        # back a batch at a time.  Workers send ONE message per batch: with a message per partition the parent's queue loop --
        # 30-50 us per item -- was what an eight-worker run waited for: 8 M reads took 1.06 s with eight workers and 0.91 s
        # with one, tools/host_ceiling.py)
        nonlocal done_count
        for _ in range(n):
            if done_count % step == 0:
                print("[freddie_segment] Done with {}/{} tints ({:.1%})".format(done_count, total, done_count / total))
            done_count += 1

    timing = os.environ.get("FREDDIE_TIMING") == "1"
    if n_gpus == 1:
        t_disc = time.perf_counter()
        jobs = [((split_dir, args.outdir, parts[i][0], parts[i][1]), parts[i][2]) for i in assign[0]]
        with boot:
            ok = False
            try:
                run_batches(make_batches(jobs, batch_bytes), params, ctx_future, args.threads, report, args.sidecar)
                ok = True
            finally:
                t_run = time.perf_counter()
                if not (ok and leave_contexts):
                    for ctx in ctx_future.result():
                        ctx.close()
                else:
                    for ctx in ctx_future.result():
                        ctx.sync()                   # (every batch's results have been written: nothing is in flight)
        if timing:
            print("[freddie_segment] discover %.3f s, batches (incl. context start-up) %.3f s, close %.3f s" % (
                t_disc - t_start, t_run - t_disc, time.perf_counter() - t_run), file=sys.stderr)
        return
    t_disc = time.perf_counter()
    for w, conn in enumerate(conns):
        conn.send((split_dir, args.outdir, [parts[i] for i in assign[w]]))
        conn.close()
    t_sent = time.perf_counter()
    t_first = None
    import queue as queue_mod
    alive = n_gpus
    while alive:
        try:
            item = queue.get(timeout=1.0)
        except queue_mod.Empty:
            dead = [pr for pr in procs if pr.exitcode not in (None, 0)]
            if dead:                                   # a worker died without reporting: stop the others, fail loudly
                for pr in procs:
                    if pr.is_alive():
                        pr.terminate()
                raise SystemExit("a GPU worker failed with exit code %s" % dead[0].exitcode)
            continue
        if item is None:
            alive -= 1
        else:
            if t_first is None:
                t_first = time.perf_counter()
            report(item)
    t_done = time.perf_counter()
    for pr in procs:
        pr.join()
        if pr.exitcode != 0:
            raise SystemExit("a GPU worker failed with exit code %s" % pr.exitcode)
    if timing:
        print("[freddie_segment] %d workers started %.3f s after main() began; discovery + scatter of %d partitions until %.3f s; job lists sent "
              "%.3f s; first partition done %.3f s; last %.3f s; workers joined %.3f s" % (
                  n_gpus, t_started - t_start, total, t_disc - t_start, t_sent - t_start, (t_first or t_done) - t_start, t_done - t_start,
                  time.perf_counter() - t_start), file=sys.stderr)


if __name__ == "__main__":
    main()
