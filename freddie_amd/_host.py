"""ctypes binding of include/freddie_host.h (native host I/O of the segmentation stage)."""
import ctypes
import os

import numpy as np

from . import build as _build

_lib = None


class HostError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    # FHOST_LIB: another build of the same library (the sanitizer build of tests/test_sanitizers.py)
    path = os.environ.get("FHOST_LIB") or _build.HOST_SO
    if not os.path.exists(path):
        raise HostError("%s not found: build it first (freddie_amd.build.build_host())" % path)
    L = ctypes.CDLL(path)
    vp = ctypes.c_void_p
    cpp = ctypes.POINTER(ctypes.c_char_p)
    L.fhost_load.restype = vp
    L.fhost_load.argtypes = [cpp, cpp, ctypes.c_int32, ctypes.c_int32]
    L.fhost_free.restype = None
    L.fhost_free.argtypes = [vp]
    L.fhost_error.restype = ctypes.c_char_p
    L.fhost_error.argtypes = [vp]
    L.fhost_n_part.restype = ctypes.c_int32
    L.fhost_n_part.argtypes = [vp]
    L.fhost_n_reads.restype = ctypes.c_int64
    L.fhost_n_reads.argtypes = [vp]
    for name in ("fhost_part_iv_off", "fhost_iv_start", "fhost_iv_end", "fhost_part_rep_off", "fhost_rep_weight",
                 "fhost_rep_exon_off", "fhost_ex_ts", "fhost_ex_te"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp]
    L.fhost_load_sidecar.restype = vp
    L.fhost_load_sidecar.argtypes = [cpp, cpp, cpp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    L.fhost_n_from_sidecar.restype = ctypes.c_int32
    L.fhost_n_from_sidecar.argtypes = [vp]
    L.fhost_sidecar_write.restype = ctypes.c_int32
    L.fhost_sidecar_write.argtypes = [vp, cpp, cpp, cpp, ctypes.c_int32]
    for name in ("fhost_write", "fhost_write_packed"):
        getattr(L, name).restype = ctypes.c_int32
        getattr(L, name).argtypes = [vp, vp, vp, vp, vp, cpp, ctypes.c_int32]
    L.fhost_discover.restype = vp
    L.fhost_discover.argtypes = [ctypes.c_char_p, ctypes.c_int32]
    L.fhost_listing_free.restype = None
    L.fhost_listing_free.argtypes = [vp]
    L.fhost_listing_error.restype = ctypes.c_char_p
    L.fhost_listing_error.argtypes = [vp]
    L.fhost_listing_n.restype = ctypes.c_int64
    L.fhost_listing_n.argtypes = [vp]
    L.fhost_listing_n_contigs.restype = ctypes.c_int32
    L.fhost_listing_n_contigs.argtypes = [vp]
    L.fhost_listing_contig.restype = ctypes.c_char_p
    L.fhost_listing_contig.argtypes = [vp, ctypes.c_int32]
    for name in ("fhost_listing_contig_of", "fhost_listing_tint", "fhost_listing_size"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp]
    L.fhost_touch.restype = ctypes.c_int32
    L.fhost_touch.argtypes = [cpp, ctypes.c_int32, ctypes.c_int32]
    _lib = L
    return L


def discover(split_dir, n_threads=8):
    """[(contig, tint_id, bytes of split_<contig>_<tint>.tsv)] of every partition under split_dir, in directory order (the native
    form of the listing loop of the reference's main(), py/freddie_segment.py:852-857)."""
    L = load()
    h = L.fhost_discover(split_dir.encode(), int(n_threads))
    if not h:
        raise HostError("fhost_discover: out of memory")
    try:
        err = L.fhost_listing_error(h).decode()
        if err:
            raise HostError(err)
        n = L.fhost_listing_n(h)
        contigs = [L.fhost_listing_contig(h, k).decode() for k in range(L.fhost_listing_n_contigs(h))]
        ci = _view(L.fhost_listing_contig_of(h), n, np.int32).tolist()
        tint = _view(L.fhost_listing_tint(h), n, np.int64).tolist()
        size = _view(L.fhost_listing_size(h), n, np.int64).tolist()
        return contigs, [(contigs[c], t, s) for c, t, s in zip(ci, tint, size)]
    finally:
        L.fhost_listing_free(h)


def touch(paths, n_threads=4):
    """Create (or truncate) the given files: the empty segment_*.log the reference leaves per partition (:695), a batch at a time."""
    if not paths:
        return
    if load().fhost_touch(_c_strings(paths), len(paths), int(n_threads)) != 0:
        raise HostError("cannot create %s ..." % paths[0])


def _c_strings(items):
    arr = (ctypes.c_char_p * len(items))()
    arr[:] = [s.encode() for s in items]
    return arr


def _view(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.empty(0, dtype)
    buf = (ctypes.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


class HostBatch:
    """Parsed partitions living in native memory; arrays() are views valid until close()."""

    def __init__(self, split_paths, reads_paths, n_threads=1, sidecar_paths=None, verify_checksum=True):
        """sidecar_paths: None (parse the TSVs) or one path-or-None per partition; a side-car that is missing,
        stale or damaged silently falls back to the TSVs of that partition."""
        self._L = load()
        self._paths = (list(split_paths), list(reads_paths))
        if sidecar_paths is None:
            self._h = self._L.fhost_load(_c_strings(split_paths), _c_strings(reads_paths), len(split_paths), int(n_threads))
        else:
            sc = (ctypes.c_char_p * len(sidecar_paths))()
            sc[:] = [None if s is None else s.encode() for s in sidecar_paths]
            self._h = self._L.fhost_load_sidecar(_c_strings(split_paths), _c_strings(reads_paths), sc, len(split_paths),
                                                 int(n_threads), 1 if verify_checksum else 0)
        if not self._h:
            raise HostError("fhost_load: out of memory")
        err = self._L.fhost_error(self._h).decode()
        if err:
            self.close()
            raise HostError(err)
        self.n_part = self._L.fhost_n_part(self._h)
        self.n_reads = self._L.fhost_n_reads(self._h)
        self.n_from_sidecar = self._L.fhost_n_from_sidecar(self._h)

    def write_sidecars(self, sidecar_paths, n_threads=1):
        rc = self._L.fhost_sidecar_write(self._h, _c_strings(self._paths[0]), _c_strings(self._paths[1]),
                                         _c_strings(sidecar_paths), int(n_threads))
        if rc != 0:
            raise HostError(self._L.fhost_error(self._h).decode())

    def arrays(self):
        L, h = self._L, self._h
        np_ = self.n_part
        part_iv_off = _view(L.fhost_part_iv_off(h), np_ + 1, np.int64)
        part_rep_off = _view(L.fhost_part_rep_off(h), np_ + 1, np.int64)
        K, R = int(part_iv_off[-1]), int(part_rep_off[-1])
        rep_exon_off = _view(L.fhost_rep_exon_off(h), R + 1, np.int64)
        I = int(rep_exon_off[-1])
        return dict(part_iv_off=part_iv_off, iv_start=_view(L.fhost_iv_start(h), K, np.int32),
                    iv_end=_view(L.fhost_iv_end(h), K, np.int32), part_rep_off=part_rep_off,
                    rep_weight=_view(L.fhost_rep_weight(h), R, np.int32), rep_exon_off=rep_exon_off,
                    ex_ts=_view(L.fhost_ex_ts(h), I, np.int32), ex_te=_view(L.fhost_ex_te(h), I, np.int32))

    def write(self, part_final_off, final_pos, label_off, labels, out_paths, n_threads=1, packed=False):
        """packed: ``labels`` holds two bits per label (Context.results(packed=True))."""
        a = [np.ascontiguousarray(part_final_off, np.int64), np.ascontiguousarray(final_pos, np.int32),
             np.ascontiguousarray(label_off, np.int64), np.ascontiguousarray(labels, np.uint8)]
        rc = (self._L.fhost_write_packed if packed else self._L.fhost_write)(self._h, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data,
                                 a[3].ctypes.data if a[3].size else None, _c_strings(out_paths), int(n_threads))
        if rc != 0:
            raise HostError(self._L.fhost_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.fhost_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
