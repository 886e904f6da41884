"""ctypes binding of include/freddie_seg.h (the C-ABI of the HIP library).

There is no CPU fallback: if the shared library is missing this raises, and if no GPU is
present ``Context()`` raises with the library's message.
"""
import ctypes
import os

import numpy as np

from . import build as _build

_lib = None


class SegError(RuntimeError):
    """code: the library's FSEG_ERR_* value (include/freddie_seg.h), None when raised by the binding itself."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


ERR_ARG, ERR_HIP, ERR_INPUT, ERR_UNSUPPORTED = 1, 2, 3, 4


class _Params(ctypes.Structure):
    _fields_ = [
        ("sigma", ctypes.c_double), ("threshold_rate", ctypes.c_double), ("variance_factor", ctypes.c_double),
        ("max_problem_size", ctypes.c_int32), ("min_read_support_outside", ctypes.c_int32),
        ("ignore_ends", ctypes.c_int32),
        ("radius_main", ctypes.c_int32), ("w_main", ctypes.c_void_p),
        ("radius_refine", ctypes.c_int32), ("w_refine", ctypes.c_void_p),
        ("h_len", ctypes.c_int32), ("h_table", ctypes.c_void_p),
    ]


class _Batch(ctypes.Structure):
    _fields_ = [
        ("n_part", ctypes.c_int32),
        ("part_iv_off", ctypes.c_void_p), ("iv_start", ctypes.c_void_p), ("iv_end", ctypes.c_void_p),
        ("part_rep_off", ctypes.c_void_p), ("rep_weight", ctypes.c_void_p), ("rep_exon_off", ctypes.c_void_p),
        ("ex_ts", ctypes.c_void_p), ("ex_te", ctypes.c_void_p),
    ]


class _Sizes(ctypes.Structure):
    _fields_ = [("n_final", ctypes.c_int64), ("label_bytes", ctypes.c_int64), ("n_cand", ctypes.c_int64),
                ("n_problems", ctypes.c_int64), ("n_positions", ctypes.c_int64), ("max_problem_size", ctypes.c_int64),
                ("max_problem_reads", ctypes.c_int64)]


EXPORTS = ["fseg_abi_version", "fseg_source_hash", "fseg_results", "fseg_results_packed", "fseg_create", "fseg_destroy", "fseg_last_error", "fseg_set_params", "fseg_upload",
           "fseg_run", "fseg_sync", "fseg_get_sizes", "fseg_download", "fseg_tap", "fseg_set_profiling",
           "fseg_n_stages", "fseg_stage_name", "fseg_stage_ms", "fseg_scoring_algorithmic_bytes"]

TAPS = dict(pos_off=(1, np.int64), y_raw=(2, np.int32), y=(3, np.float64), threshold=(4, np.float64),
            cand_off=(5, np.int64), cand_y=(6, np.int32), fixed=(7, np.uint8), chosen=(8, np.uint8),
            final_off=(9, np.int64), final_y=(10, np.int32), problems=(11, np.int32),
            lane_start=(12, np.int32), lane_pmax=(13, np.int32), lane_exons=(14, np.int64),
            lane_stream=(15, np.int32), exon_stream=(16, np.int32), sync=(17, np.int32))


def lib_path():
    # FSEG_LIB: developer override used to compare builds of the same library (tuning experiments)
    return os.environ.get("FSEG_LIB") or _build.SEG_SO


def load():
    """Load libfreddie_seg.so (must have been built in-tree: ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise SegError("%s not found: build it first (freddie_amd.build.build_seg()); there is no CPU fallback" % path)
    L = ctypes.CDLL(path)
    vp = ctypes.c_void_p
    L.fseg_abi_version.restype = ctypes.c_int
    L.fseg_create.restype = ctypes.c_int
    L.fseg_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.fseg_destroy.restype = None
    L.fseg_destroy.argtypes = [vp]
    L.fseg_last_error.restype = ctypes.c_char_p
    L.fseg_last_error.argtypes = [vp]
    L.fseg_set_params.restype = ctypes.c_int
    L.fseg_set_params.argtypes = [vp, ctypes.POINTER(_Params)]
    L.fseg_upload.restype = ctypes.c_int
    L.fseg_upload.argtypes = [vp, ctypes.POINTER(_Batch)]
    for n in ("fseg_run", "fseg_sync"):
        getattr(L, n).restype = ctypes.c_int
        getattr(L, n).argtypes = [vp]
    L.fseg_get_sizes.restype = ctypes.c_int
    L.fseg_get_sizes.argtypes = [vp, ctypes.POINTER(_Sizes)]
    L.fseg_download.restype = ctypes.c_int
    L.fseg_download.argtypes = [vp, vp, vp, vp, vp]
    L.fseg_results.restype = ctypes.c_int
    L.fseg_results.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.fseg_results_packed.restype = ctypes.c_int
    L.fseg_results_packed.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.fseg_source_hash.restype = ctypes.c_char_p
    L.fseg_tap.restype = ctypes.c_int
    L.fseg_tap.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    L.fseg_set_profiling.restype = ctypes.c_int
    L.fseg_set_profiling.argtypes = [vp, ctypes.c_int]
    L.fseg_n_stages.restype = ctypes.c_int
    L.fseg_stage_name.restype = ctypes.c_char_p
    L.fseg_stage_name.argtypes = [ctypes.c_int]
    L.fseg_stage_ms.restype = ctypes.c_int
    L.fseg_stage_ms.argtypes = [vp, vp]
    L.fseg_scoring_algorithmic_bytes.restype = ctypes.c_int64
    L.fseg_scoring_algorithmic_bytes.argtypes = [vp]
    if L.fseg_abi_version() != 2:
        raise SegError("libfreddie_seg.so ABI version mismatch")
    if not os.environ.get("FSEG_LIB"):
        # the built library is git-ignored and travels with the tree: refuse one that was built from other sources
        want = _build.seg_hash()
        have = L.fseg_source_hash().decode()
        if have != want:
            raise SegError("%s is stale (built from sources %s, the tree is %s): rebuild with "
                           "python -c 'import __graft_entry__ as g; g.build()'" % (path, have or "?", want))
    _lib = L
    return L


class Context:
    """One GPU context (one per device; not shared between threads)."""

    def __init__(self, device=0, handle=None):
        """``handle``: a context fseg_create() has already made for this device (freddie_amd/_early.py)."""
        self._L = load()
        h = handle if handle is not None else ctypes.c_void_p()
        if handle is None:
            rc = self._L.fseg_create(int(device), ctypes.byref(h))
            if rc != 0:
                raise SegError("fseg_create(device=%d) failed: %s" % (device, self._L.fseg_last_error(None).decode()))
        self._h = h
        self._keep = []
        self.n_part = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.fseg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise SegError("%s failed (%d): %s" % (what, rc, self._L.fseg_last_error(self._h).decode()), rc)

    def set_params(self, sigma, threshold_rate, variance_factor, max_problem_size, min_read_support_outside,
                   ignore_ends, w_main, w_refine, h_table):
        w_main = np.ascontiguousarray(w_main, np.float64)
        w_refine = np.ascontiguousarray(w_refine, np.float64)
        h_table = np.ascontiguousarray(h_table, np.float64)
        p = _Params(float(sigma), float(threshold_rate), float(variance_factor), int(max_problem_size),
                    int(min_read_support_outside), 1 if ignore_ends else 0, len(w_main) - 1, w_main.ctypes.data,
                    len(w_refine) - 1, w_refine.ctypes.data, len(h_table), h_table.ctypes.data)
        self._check(self._L.fseg_set_params(self._h, ctypes.byref(p)), "fseg_set_params")

    def upload(self, part_iv_off, iv_start, iv_end, part_rep_off, rep_weight, rep_exon_off, ex_ts, ex_te):
        a = [np.ascontiguousarray(part_iv_off, np.int64), np.ascontiguousarray(iv_start, np.int32),
             np.ascontiguousarray(iv_end, np.int32), np.ascontiguousarray(part_rep_off, np.int64),
             np.ascontiguousarray(rep_weight, np.int32), np.ascontiguousarray(rep_exon_off, np.int64),
             np.ascontiguousarray(ex_ts, np.int32), np.ascontiguousarray(ex_te, np.int32)]
        n_part = len(a[0]) - 1
        if len(a[3]) != n_part + 1 or len(a[1]) != a[0][-1] or len(a[2]) != a[0][-1] or len(a[4]) != a[3][-1] \
                or len(a[5]) != a[3][-1] + 1 or len(a[6]) != a[5][-1] or len(a[7]) != a[5][-1]:
            raise SegError("upload: array lengths do not match the offsets")
        b = _Batch(n_part, *[x.ctypes.data for x in a])
        self._check(self._L.fseg_upload(self._h, ctypes.byref(b)), "fseg_upload")
        self.n_part = n_part
        self._rep_counts = np.diff(a[3])

    def run(self):
        self._check(self._L.fseg_run(self._h), "fseg_run")

    def sync(self):
        self._check(self._L.fseg_sync(self._h), "fseg_sync")

    def sizes(self):
        s = _Sizes()
        self._check(self._L.fseg_get_sizes(self._h, ctypes.byref(s)), "fseg_get_sizes")
        return {k: getattr(s, k) for k, _ in _Sizes._fields_}

    def download(self, labels=True):
        """Returns (part_final_off int64[n_part+1], final_pos int32[F], label_off int64[n_part+1], labels uint8[...])."""
        sz = self.sizes()
        pfo = np.empty(self.n_part + 1, np.int64)
        fp = np.empty(sz["n_final"], np.int32)
        lo = np.empty(self.n_part + 1, np.int64)
        lb = np.empty(sz["label_bytes"] if labels else 0, np.uint8)
        self._check(self._L.fseg_download(self._h, pfo.ctypes.data, fp.ctypes.data, lo.ctypes.data,
                                          lb.ctypes.data if labels and lb.size else None), "fseg_download")
        return pfo, fp, lo, lb

    def results(self, packed=False):
        """Like download(), without the copy: numpy views of the context's pinned result buffers, valid until the next
        run / upload / results() on this context (what the pipelined CLI hands to the native writer).
        packed=True: the labels at two bits each (label g = bits 2(g & 3).. of byte g >> 2), a quarter of the PCIe traffic;
        HostBatch.write(..., packed=True) takes them."""
        p = [ctypes.c_void_p() for _ in range(4)]
        fn = self._L.fseg_results_packed if packed else self._L.fseg_results
        self._check(fn(self._h, *[ctypes.byref(x) for x in p]), "fseg_results")
        sz = self.sizes()

        def view(ptr, n, dtype):
            if n == 0 or not ptr.value:
                return np.empty(0, dtype)
            buf = (ctypes.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=dtype)
        return (view(p[0], self.n_part + 1, np.int64), view(p[1], sz["n_final"], np.int32),
                view(p[2], self.n_part + 1, np.int64), view(p[3], (sz["label_bytes"] + 3) // 4 if packed else sz["label_bytes"], np.uint8))

    def tap(self, name):
        what, dtype = TAPS[name]
        n = ctypes.c_int64()
        self._check(self._L.fseg_tap(self._h, what, None, 0, ctypes.byref(n)), "fseg_tap")
        out = np.empty(n.value // np.dtype(dtype).itemsize, dtype)
        if n.value:
            self._check(self._L.fseg_tap(self._h, what, out.ctypes.data, n.value, ctypes.byref(n)), "fseg_tap")
        return out.reshape(-1, 4) if name == "problems" else (out.reshape(-1, 2) if name in ("lane_exons", "lane_stream", "exon_stream") else out)

    def set_profiling(self, on):
        """True / 1: HIP events around every stage (first runs; replays report the scoring stage and the two graphs around it);
        2: around the interval-scoring stage only; 3: around every stage on replays too (plain launches instead of the graph);
        False: none."""
        self._check(self._L.fseg_set_profiling(self._h, on if on in (2, 3) else (1 if on else 0)), "fseg_set_profiling")

    def stage_ms(self):
        n = self._L.fseg_n_stages()
        ms = np.zeros(n, np.float32)
        self._check(self._L.fseg_stage_ms(self._h, ms.ctypes.data), "fseg_stage_ms")
        return {self._L.fseg_stage_name(i).decode(): float(ms[i]) for i in range(n)}

    def scoring_algorithmic_bytes(self):
        return int(self._L.fseg_scoring_algorithmic_bytes(self._h))
