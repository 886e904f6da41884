"""Pre-ILP work of the clustering stage: the host-side mirror of the reference's
``py/freddie_cluster.py`` up to (not including) the ILP, with the two quadratic loops of ``partition_reads`` done by
the gfx950 library behind ``include/freddie_cluster.h``.

Reference map (file:line of vpc-ccg/freddie ``py/freddie_cluster.py``):
  read_segment :119-172        -> read_segment()          segment_*.tsv -> tint dicts (reads, read_reps by structure key)
  find_segment_read :175-183   -> find_segment_read()
  preprocess_ilp :277-328      -> preprocess_ilp()        I / C / FL / garbage_cost, poly-tail categories
  split_list_evenly :112-116   -> split_list_evenly()
  partition_reads :196-274     -> partition_reads(), partition_reads_batch()
       unique structures :203-215 (host), pairwise compatibility :217-234 and edge pruning :240-255 (GPU),
       connected components :256-257, even split and incompatible pairs :258-274 (host)
The ILP (run_ilp, Gurobi) and everything after it are out of scope.  There is no CPU implementation of the two
quadratic loops in this package: without the HIP library partition_reads() raises.
"""
import ctypes
import os
import re
from math import ceil

import numpy as np

from . import build as _build

# ---------------------------------------------------------------------------------------------------------------
# segment_*.tsv -> tint dict   (read_segment :119-172)
# ---------------------------------------------------------------------------------------------------------------
_CHR = r"[0-9A-Za-z!#$%&+./:;?@^_|~-][0-9A-Za-z!#$%&*+./:;=?@^_|~-]*"
_HEADER = re.compile(r"#(" + _CHR + r")\t([0-9]+)\t([0-9]+(?:,[0-9]+)*)\n$")
_INTERNAL = r"(\d+)-(\d+):(\d+),"
_SOFTCLIP = r"([ES]SC):(\d+),"
_POLY = r"([ES][AT])_(\d+):(\d+),"
_READ = re.compile(r"([0-9]+)\t([!-?A-~]{1,254})\t(" + _CHR + r")\t([+-])\t([0-9]+)\t([012]+)\t((?:" + _INTERNAL + "|" +
                   _SOFTCLIP + "|" + _POLY + r")*)\n$")
_INTERNAL_RE, _SOFTCLIP_RE, _POLY_RE = re.compile(_INTERNAL), re.compile(_SOFTCLIP), re.compile(_POLY)


def read_segment(segment_tsv):
    """{tint id: tint}; tint = id, chr, segs [(start, end, length)], reads [...], read_reps [[read index, ...], ...]
    where reads with the same structure key (labels with 2 -> 0, large internal gaps, large poly tails) share a rep."""
    tints = dict()
    keys = dict()
    for line in open(segment_tsv):
        if line[0] == "#":
            m = _HEADER.match(line)
            pos = [int(x) for x in m.group(3).split(",")]
            assert all(a < b for a, b in zip(pos[:-1], pos[1:])), pos
            tid = int(m.group(2))
            assert tid not in tints, "Transcriptional interval with id {} is repeated!".format(tid)
            tints[tid] = dict(id=tid, chr=m.group(1), segs=[(s, e, e - s) for s, e in zip(pos[:-1], pos[1:])],
                              read_reps=list(), reads=list())
            keys[tid] = dict()
            continue
        m = _READ.match(line)
        rid, name, chrom, strand, cid, data, gaps = m.group(1, 2, 3, 4, 5, 6, 7)
        internal = _INTERNAL_RE.findall(gaps)
        poly = _POLY_RE.findall(gaps)
        read = dict(id=int(rid), name=name, chr=chrom, strand=strand, tint=int(cid), data=[int(d) for d in data],
                    gaps={(int(a), int(b)): int(c) for a, b, c in internal},
                    softclip={k: int(v) for k, v in _SOFTCLIP_RE.findall(gaps)},
                    poly_tail={k: (int(a), int(b)) for k, a, b in poly})
        key = data.replace("2", "0")
        key += "".join(".{}".format(c if int(c) > 10 else 0) for _, _, c in internal)
        key += "".join(".{}{}".format(k[0], b if int(b) > 10 else 0) for k, _, b in poly)
        tint = tints[read["tint"]]
        tint["reads"].append(read)
        reps = keys[read["tint"]]
        if key not in reps:
            reps[key] = len(tint["read_reps"])
            tint["read_reps"].append(list())
        tint["read_reps"][reps[key]].append(len(tint["reads"]) - 1)
        assert len(read["data"]) == len(tint["segs"]), (read["data"], tint["segs"])
        assert read["chr"] == tint["chr"]
        assert all(0 <= a < b < len(read["data"]) for a, b in read["gaps"].keys())
    return tints


def find_segment_read(M, i):
    """(first, last) segment of row i holding a 1; (-1, len - 1) when there is none (:175-183)."""
    row = M[i]
    first, last = -1, len(row) - 1
    for j, v in enumerate(row):
        if v == 1:
            if first == -1:
                first = j
            last = j
    return (first, last)


def garbage_cost_introns(C):
    return max(sum(C.values()) - 0.5, 1)


def garbage_cost_exons(I):
    return max(sum(I.values()) - 0.5, 1)


def preprocess_ilp(tint, ilp_settings):
    """tint['ilp_data'] = I (label % 2), C (0-labels strictly inside the read's span), FL (first, last), garbage_cost;
    sets poly_tail_category on every read and the tail pseudo-gaps (-1, first) / (last, M) on the reps (:277-328)."""
    read_reps = tint["read_reps"]
    M = len(tint["segs"])
    I, C, FL = dict(), dict(), dict()
    for i, members in enumerate(read_reps):
        read = tint["reads"][members[0]]
        I[i] = [v % 2 for v in read["data"][:M]]
        C[i] = [0] * M
        lo, hi = find_segment_read(I, i)
        read["poly_tail_category"] = "N"
        if len(read["poly_tail"]) == 1:
            key = next(iter(read["poly_tail"]))
            length, gap = read["poly_tail"][key]
            if key in ("SA", "ST") and length > 10:
                read["poly_tail_category"] = "S"
                read["gaps"][(-1, lo)] = gap
                lo = 0
            elif key in ("EA", "ET") and length > 10:
                read["poly_tail_category"] = "E"
                read["gaps"][(hi, M)] = gap
                hi = M - 1
        FL[i] = (lo, hi)
        for j in range(M):
            C[i][j] = 1 if (lo <= j <= hi and read["data"][j] == 0) else 0
        for ridx in members:
            tint["reads"][ridx]["poly_tail_category"] = read["poly_tail_category"]
            tint["reads"][ridx]["gaps"] = read["gaps"]
    garbage_cost = {}
    for i in range(len(read_reps)):
        if ilp_settings["recycle_model"] == "exons":
            garbage_cost[i] = len(read_reps[i]) * garbage_cost_exons(I=I[i])      # a list: raises like the reference (:314)
        elif ilp_settings["recycle_model"] == "introns":
            garbage_cost[i] = len(read_reps[i]) * garbage_cost_introns(C=C[i])  # a list: raises like the reference (:316)
        elif ilp_settings["recycle_model"] == "constant":
            garbage_cost[i] = len(read_reps[i]) * 3
    tint["ilp_data"] = dict(FL=FL, I=I, C=C, garbage_cost=garbage_cost)


def split_list_evenly(l, m):
    p = ceil(len(l) / m)
    s = ceil(len(l) / p)
    for idx in range(0, p * s, s):
        yield l[idx:idx + s]


# ---------------------------------------------------------------------------------------------------------------
# the C-ABI (include/freddie_cluster.h)
# ---------------------------------------------------------------------------------------------------------------
CLUSTER_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfreddie_cluster.so")
CLUSTER_SRC = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "freddie_cluster.hip")]
EXPORTS = ["fclu_abi_version", "fclu_create", "fclu_destroy", "fclu_last_error", "fclu_compat_graph", "fclu_last_timing"]
_lib = None


class ClusterError(RuntimeError):
    pass


class _Batch(ctypes.Structure):
    _fields_ = [("n_tint", ctypes.c_int32), ("row_off", ctypes.c_void_p), ("n_seg", ctypes.c_void_p),
                ("bits_off", ctypes.c_void_p), ("bits", ctypes.c_void_p), ("first", ctypes.c_void_p),
                ("last", ctypes.c_void_p), ("tail", ctypes.c_void_p), ("adj_off", ctypes.c_void_p)]


def build(force=False, verbose=False):
    """hipcc -> libfreddie_cluster.so (in-tree; cross-compiles without a GPU)."""
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-I", _build.INCLUDE, "-o", CLUSTER_SO] + CLUSTER_SRC
    _build.build_stamped(CLUSTER_SO, cmd, CLUSTER_SRC + [os.path.join(_build.INCLUDE, "freddie_cluster.h")], force, verbose)
    return CLUSTER_SO


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(CLUSTER_SO):
        raise ClusterError("%s not found: build it first (freddie_amd.cluster_prep.build()); there is no CPU fallback" % CLUSTER_SO)
    L = ctypes.CDLL(CLUSTER_SO)
    vp = ctypes.c_void_p
    L.fclu_abi_version.restype = ctypes.c_int
    L.fclu_create.restype = ctypes.c_int
    L.fclu_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.fclu_destroy.restype = None
    L.fclu_destroy.argtypes = [vp]
    L.fclu_last_error.restype = ctypes.c_char_p
    L.fclu_last_error.argtypes = [vp]
    L.fclu_compat_graph.restype = ctypes.c_int
    L.fclu_compat_graph.argtypes = [vp, ctypes.POINTER(_Batch), ctypes.c_int32, vp, vp]
    L.fclu_last_timing.restype = ctypes.c_int
    L.fclu_last_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    _lib = L
    return L


class Context:
    """One GPU context of the clustering pre-ILP library."""

    def __init__(self, device=0):
        self._L = load()
        h = ctypes.c_void_p()
        rc = self._L.fclu_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise ClusterError("fclu_create: " + self._L.fclu_last_error(None).decode())
        self._h = h

    def compat_graph(self, packed, prune=True):
        """packed: pack_structures() of a batch.  Returns (adj uint64[adj_off[-1]], rounds int32[n_tint])."""
        b = _Batch(n_tint=packed["n_tint"])
        keep = []
        for name, dt in (("row_off", np.int64), ("n_seg", np.int32), ("bits_off", np.int64), ("bits", np.uint32),
                         ("first", np.int32), ("last", np.int32), ("tail", np.uint8), ("adj_off", np.int64)):
            a = np.ascontiguousarray(packed[name], dt)
            keep.append(a)
            setattr(b, name, a.ctypes.data if a.size else None)
        adj = np.zeros(max(int(packed["adj_off"][-1]), 1), np.uint64)
        rounds = np.zeros(packed["n_tint"], np.int32)
        rc = self._L.fclu_compat_graph(self._h, ctypes.byref(b), 1 if prune else 0, adj.ctypes.data, rounds.ctypes.data)
        if rc != 0:
            raise ClusterError("fclu_compat_graph: " + self._L.fclu_last_error(self._h).decode())
        return adj[:int(packed["adj_off"][-1])], rounds

    def last_timing(self):
        a, b = ctypes.c_float(), ctypes.c_float()
        self._L.fclu_last_timing(self._h, ctypes.byref(a), ctypes.byref(b))
        return dict(compat_ms=a.value, prune_ms=b.value)

    def close(self):
        if getattr(self, "_h", None):
            self._L.fclu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------------
# partition_reads (:196-274)
# ---------------------------------------------------------------------------------------------------------------
_TAIL_CODE = {"N": 0, "S": 1, "E": 2}


def unique_structures(tint):
    """[(structure, [rep ids])] in first-occurrence order: reps with the same I row, first/last and tail category (:203-215)."""
    reads, read_reps = tint["reads"], tint["read_reps"]
    I, FL = tint["ilp_data"]["I"], tint["ilp_data"]["FL"]
    seen = dict()
    for i in sorted(I.keys()):
        d = (tuple(I[i]), (FL[i][0], FL[i][1], reads[read_reps[i][0]]["poly_tail_category"]))
        seen.setdefault(d, []).append(i)
    return list(seen.items())


def pack_structures(unique_per_tint):
    """Flat arrays of include/freddie_cluster.h for a list of unique_structures() results."""
    T = len(unique_per_tint)
    row_off = np.zeros(T + 1, np.int64); bits_off = np.zeros(T + 1, np.int64); adj_off = np.zeros(T + 1, np.int64)
    n_seg = np.zeros(T, np.int32)
    bits, first, last, tail = [], [], [], []
    for t, uniq in enumerate(unique_per_tint):
        n = len(uniq)
        M = len(uniq[0][0][0]) if n else 0
        W = max((M + 31) // 32, 1)
        n_seg[t] = M
        row_off[t + 1] = row_off[t] + n
        bits_off[t + 1] = bits_off[t] + n * W
        adj_off[t + 1] = adj_off[t] + n * ((n + 63) // 64)
        if n:
            rows = np.zeros((n, W * 32), np.uint8)
            if M:
                rows[:, :M] = np.array([u[0][0] for u in uniq], np.uint8).reshape(n, M)
            bits.append(np.packbits(rows, axis=1, bitorder="little").view(np.uint32).reshape(-1))
            first.extend(u[0][1][0] for u in uniq)
            last.extend(u[0][1][1] for u in uniq)
            tail.extend(_TAIL_CODE[u[0][1][2]] for u in uniq)
    return dict(n_tint=T, row_off=row_off, n_seg=n_seg, bits_off=bits_off,
                bits=np.concatenate(bits) if bits else np.zeros(0, np.uint32), first=np.array(first, np.int32),
                last=np.array(last, np.int32), tail=np.array(tail, np.uint8), adj_off=adj_off)


def adjacency_matrix(adj, packed, t):
    """Boolean N_t x N_t matrix of tint t from the packed result."""
    n = int(packed["row_off"][t + 1] - packed["row_off"][t])
    aw = (n + 63) // 64
    words = adj[int(packed["adj_off"][t]):int(packed["adj_off"][t + 1])].reshape(n, aw) if n else np.zeros((0, 0), np.uint64)
    return np.unpackbits(words.view(np.uint8), axis=1, bitorder="little")[:, :n].astype(bool) if n else np.zeros((0, 0), bool)


def _components(A):
    """Connected components as sorted lists, ordered by their smallest node (networkx yields them in node order, :257)."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components
    n = A.shape[0]
    if n == 0:
        return []
    _, lab = connected_components(csr_matrix(A), directed=False)
    order = np.argsort(lab, kind="stable")
    groups = np.split(order, np.flatnonzero(np.diff(lab[order])) + 1)
    return sorted((g.tolist() for g in groups), key=lambda g: g[0])


def _partitions_from_graph(unique, A, maximum_ilp_size, verbose):
    parts = []
    for comp in _components(A):
        for c in split_list_evenly(comp, maximum_ilp_size):
            if verbose:
                print(len(c), c[:10])                       # the reference prints this line (:262)
            rids, incomp = [], []
            for idx, i in enumerate(c):
                rids.extend(unique[i][1])
                for j in c[idx + 1:]:
                    if A[i, j]:
                        continue
                    for rid_1 in unique[i][1]:
                        for rid_2 in unique[j][1]:
                            incomp.append((rid_1, rid_2))
            parts.append((rids, incomp))
    return parts


def partition_reads_batch(tints, maximum_ilp_size, ctx, verbose=True):
    """partition_reads() of several preprocessed tints with one device call; sets tint['partitions'] on each."""
    uniq = [unique_structures(t) for t in tints]
    packed = pack_structures(uniq)
    adj, _ = ctx.compat_graph(packed, prune=True)
    for t, tint in enumerate(tints):
        tint["partitions"] = _partitions_from_graph(uniq[t], adjacency_matrix(adj, packed, t), maximum_ilp_size, verbose)


def partition_reads(tint, maximum_ilp_size, ctx=None, verbose=True):
    """Reference-shaped entry (:196): tint['partitions'] = [(rep ids, [(incompatible rep pair), ...]), ...]."""
    own = ctx is None
    ctx = ctx or Context(0)
    try:
        partition_reads_batch([tint], maximum_ilp_size, ctx, verbose)
    finally:
        if own:
            ctx.close()
