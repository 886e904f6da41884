"""ctypes front-end of the C oracle (TEST INFRASTRUCTURE -- see freddie_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfreddie_oracle.so")


def build(force=False):
    # rebuilt when the library does not carry the hash of its source + Makefile (same scheme as freddie_amd/build.py;
    # restated here because the oracle imports nothing from the product)
    import hashlib
    h = hashlib.sha256()
    for name in ("Makefile", "freddie_oracle.c"):
        with open(os.path.join(_HERE, name), "rb") as f:
            h.update(f.read())
    want = h.hexdigest()[:32]
    have = None
    if os.path.exists(_SO):
        with open(_SO, "rb") as f:
            blob = f.read()
        i = blob.find(b"FREDDIE_SRC_HASH=")
        have = blob[i + 17:i + 49].decode("ascii", "replace") if i >= 0 else None
    if force or have != want:
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "libfreddie_oracle.so", 'STAMP=-DFREDDIE_SOURCE_HASH=\\"%s\\"' % want])
    return _SO


class _Params(ctypes.Structure):
    _fields_ = [
        ("sigma", ctypes.c_double), ("threshold_rate", ctypes.c_double), ("variance_factor", ctypes.c_double),
        ("max_problem_size", ctypes.c_int32), ("min_read_support_outside", ctypes.c_int32),
        ("ignore_ends", ctypes.c_int32),
        ("radius_main", ctypes.c_int32), ("w_main", ctypes.c_void_p),
        ("radius_refine", ctypes.c_int32), ("w_refine", ctypes.c_void_p),
        ("h_len", ctypes.c_int32), ("h_table", ctypes.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(os.environ.get("FREDDIE_ORACLE_SO") or _SO)      # override: the sanitizer build (tests/test_sanitizers.py)
        vp = ctypes.c_void_p
        L.fo_segment.restype = vp
        L.fo_segment.argtypes = [ctypes.POINTER(_Params), ctypes.c_int32, vp, vp, ctypes.c_int32, vp, vp, vp, vp, ctypes.c_int32]
        L.fo_free.argtypes = [vp]
        for n in ("fo_P", "fo_F", "fo_n_problems", "fo_n_vals"):
            getattr(L, n).restype = ctypes.c_int64; getattr(L, n).argtypes = [vp]
        L.fo_threshold.restype = ctypes.c_double; L.fo_threshold.argtypes = [vp]
        L.fo_error.restype = ctypes.c_int32; L.fo_error.argtypes = [vp]
        L.fo_errmsg.restype = ctypes.c_char_p; L.fo_errmsg.argtypes = [vp]
        for n in ("fo_pos_off", "fo_Y_raw", "fo_Y", "fo_cand_off", "fo_cands", "fo_fixed_off", "fo_fixed",
                  "fo_prob_interval", "fo_prob_start", "fo_prob_end", "fo_prob_nchain", "fo_finalc_off", "fo_finalc",
                  "fo_refine_off", "fo_refine", "fo_final_off", "fo_final_y", "fo_final_pos", "fo_labels"):
            getattr(L, n).restype = vp; getattr(L, n).argtypes = [vp]
        L.fo_gaussian.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int32, ctypes.c_int32, vp]
        L.fo_np_sum.restype = ctypes.c_double; L.fo_np_sum.argtypes = [vp, ctypes.c_int64]
        L.fo_local_maxima.restype = ctypes.c_int64; L.fo_local_maxima.argtypes = [vp, ctypes.c_int64, vp]
        L.fo_select_by_distance.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int32, vp]
        L.fo_variance_threshold.restype = ctypes.c_double
        L.fo_variance_threshold.argtypes = [vp, ctypes.c_int64, ctypes.c_double, vp]
        _lib = L
    return _lib


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.empty(0, dtype)
    buf = (ctypes.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


def gaussian_half_kernel(sigma, truncate):
    """Half kernel (centre first) exactly as scipy.ndimage builds it
    (scipy/ndimage/_filters.py _gaussian_kernel1d; call sites py/freddie_segment.py:755,:260)."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    return np.ascontiguousarray(phi[radius:], dtype=np.float64)


def smooth_threshold_table(threshold):
    """py/freddie_segment.py:277-286 restated."""
    out = []
    while True:
        x = len(out)
        y = threshold / (1 + ((threshold - .5) / .5) * math.exp(-0.05 * x))
        if x > 5 and x * (threshold - y) < 0.5:
            break
        out.append(round(y, 2))
        if len(out) >= 1000:
            raise AssertionError("smooth_threshold does not converge")
    return np.array(out, dtype=np.float64)


def segment(iv_start, iv_end, rep_weight, rep_exon_off, ex_ts, ex_te, sigma=5.0, threshold_rate=0.9,
            variance_factor=3.0, max_problem_size=50, min_read_support_outside=3, ignore_ends=True,
            w_main=None, w_refine=None, h_table=None, stop_after=0):
    """Run the oracle on one partition given as flat arrays; returns a dict of numpy arrays."""
    L = lib()
    iv_start = np.ascontiguousarray(iv_start, np.int32); iv_end = np.ascontiguousarray(iv_end, np.int32)
    rep_weight = np.ascontiguousarray(rep_weight, np.int32)
    rep_exon_off = np.ascontiguousarray(rep_exon_off, np.int64)
    ex_ts = np.ascontiguousarray(ex_ts, np.int32); ex_te = np.ascontiguousarray(ex_te, np.int32)
    w_main = gaussian_half_kernel(sigma, 4.0) if w_main is None else np.ascontiguousarray(w_main, np.float64)
    w_refine = gaussian_half_kernel(sigma, 1.0) if w_refine is None else np.ascontiguousarray(w_refine, np.float64)
    h_table = smooth_threshold_table(threshold_rate) if h_table is None else np.ascontiguousarray(h_table, np.float64)
    p = _Params(sigma, threshold_rate, variance_factor, max_problem_size, min_read_support_outside,
                1 if ignore_ends else 0, len(w_main) - 1, w_main.ctypes.data, len(w_refine) - 1,
                w_refine.ctypes.data, len(h_table), h_table.ctypes.data)
    K, R = len(iv_start), len(rep_weight)
    h = L.fo_segment(ctypes.byref(p), K, iv_start.ctypes.data, iv_end.ctypes.data, R, rep_weight.ctypes.data,
                     rep_exon_off.ctypes.data, ex_ts.ctypes.data, ex_te.ctypes.data, stop_after)
    try:
        err = L.fo_error(h)
        out = dict(error=err, errmsg=L.fo_errmsg(h).decode() if err else "")
        P = L.fo_P(h)
        out["pos_off"] = _arr(L.fo_pos_off(h), K + 1, np.int64)
        out["Y_raw"] = _arr(L.fo_Y_raw(h), P, np.float64)
        out["Y"] = _arr(L.fo_Y(h), P, np.float64)
        out["threshold"] = L.fo_threshold(h)
        out["n_vals"] = L.fo_n_vals(h)
        if L.fo_cand_off(h):
            out["cand_off"] = _arr(L.fo_cand_off(h), K + 1, np.int64)
            out["cands"] = _arr(L.fo_cands(h), out["cand_off"][-1], np.int32)
            out["fixed_off"] = _arr(L.fo_fixed_off(h), K + 1, np.int64)
            out["fixed"] = _arr(L.fo_fixed(h), out["fixed_off"][-1], np.int32)
        if stop_after == 0 and not err:
            npb = L.fo_n_problems(h)
            out["prob_interval"] = _arr(L.fo_prob_interval(h), npb, np.int32)
            out["prob_start"] = _arr(L.fo_prob_start(h), npb, np.int32)
            out["prob_end"] = _arr(L.fo_prob_end(h), npb, np.int32)
            out["prob_nchain"] = _arr(L.fo_prob_nchain(h), npb, np.int32)
            out["finalc_off"] = _arr(L.fo_finalc_off(h), K + 1, np.int64)
            out["finalc"] = _arr(L.fo_finalc(h), out["finalc_off"][-1], np.int32)
            out["refine_off"] = _arr(L.fo_refine_off(h), K + 1, np.int64)
            out["refine"] = _arr(L.fo_refine(h), out["refine_off"][-1], np.int32)
            out["final_off"] = _arr(L.fo_final_off(h), K + 1, np.int64)
            F = L.fo_F(h)
            out["final_y"] = _arr(L.fo_final_y(h), F, np.int32)
            out["final_pos"] = _arr(L.fo_final_pos(h), F, np.int32)
            out["labels"] = _arr(L.fo_labels(h), R * max(F - 1, 0), np.uint8).reshape(R, max(F - 1, 0))
    finally:
        L.fo_free(h)
    return out


def gaussian(x, w, mode):
    L = lib()
    x = np.ascontiguousarray(x, np.float64); w = np.ascontiguousarray(w, np.float64)
    out = np.empty_like(x)
    L.fo_gaussian(x.ctypes.data, len(x), w.ctypes.data, len(w) - 1, {"reflect": 0, "constant": 1}[mode], out.ctypes.data)
    return out


def np_sum(a):
    a = np.ascontiguousarray(a, np.float64)
    return lib().fo_np_sum(a.ctypes.data, len(a))


def variance_threshold(Y, vf):
    Y = np.ascontiguousarray(Y, np.float64)
    return lib().fo_variance_threshold(Y.ctypes.data, len(Y), float(vf), None)


def local_maxima(y):
    y = np.ascontiguousarray(y, np.float64)
    out = np.empty(max(len(y), 1), np.int32)
    m = lib().fo_local_maxima(y.ctypes.data, len(y), out.ctypes.data)
    return out[:m].copy()


def peaks_with_distance(y, distance):
    pk = local_maxima(y)
    pr = np.ascontiguousarray(np.asarray(y, np.float64)[pk])
    keep = np.empty(max(len(pk), 1), np.uint8)
    lib().fo_select_by_distance(pk.ctypes.data, pr.ctypes.data, len(pk), int(distance), keep.ctypes.data)
    return pk[keep[:len(pk)].astype(bool)]
