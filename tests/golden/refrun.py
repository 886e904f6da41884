"""Run the REFERENCE segmentation stage (imported from /root/reference) and record its
intermediates.  Only usable in the build container: the reference never travels to the GPU
box.  Used by make_golden.py (fixture minting) and by the optional live differential tests.

The reference module is imported unmodified; recording is done by wrapping its module-level
functions (looked up at call time by ``segment()``), which does not change what they compute.
"""
import importlib.util
import os
import warnings

import numpy as np

REF_PATH = "/root/reference/py/freddie_segment.py"


def available():
    return os.path.exists(REF_PATH)


_ref = None


def ref():
    global _ref
    if _ref is None:
        spec = importlib.util.spec_from_file_location("freddie_segment_reference", REF_PATH)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        _ref = m
    return _ref


def run_recorded(split_dir, outdir, contig, tint_id, sigma=5.0, threshold_rate=0.9, variance_factor=3.0,
                 max_problem_size=50, min_read_support_outside=3, consider_ends=False, allow_raise=False):
    """Runs the reference's own run_segment() (py/freddie_segment.py:681-735) on one partition,
    writing ``<outdir>/<contig>/segment_<contig>_<tint>.tsv`` with the reference's writer.
    Returns (tint, rec): the mutated tint dict and a dict of recorded intermediates.
    ``allow_raise``: an exception of the reference (its own assertions, an IndexError of ``break_large_problems`` :640) is
    recorded as ``rec["raised"]`` (type name and message) instead of propagating; the tint is the parsed input then."""
    R = ref()
    rec = dict(Y_raw=None, Y=[], cands=[], fixed=[], final_c=[], refine=[], problems=[], tint=None)
    names = ["gaussian_filter1d", "candidates_from_peaks", "break_large_problems", "run_optimize",
             "refine_segmentation", "optimize", "process_splicing_data", "segment"]
    orig = {k: getattr(R, k) for k in names}
    state = dict(in_refine=False)

    def w_psd(tint, ignore_ends):
        out = orig["process_splicing_data"](tint, ignore_ends)
        rec["Y_raw"] = [np.array(y, dtype=np.float64) for y in out[2]]
        return out

    def w_gauss(x, sigma, **kw):
        y = orig["gaussian_filter1d"](x, sigma, **kw)
        if not state["in_refine"]:
            rec["Y"].append(np.array(y, dtype=np.float64))
        return y

    def w_cands(y):
        c = orig["candidates_from_peaks"](y)
        rec["cands"].append([int(v) for v in c])
        return c

    def w_break(cands, fixed, y, mps, window=5):
        out = orig["break_large_problems"](cands, fixed, y, mps, window)
        rec["fixed"].append(sorted(int(v) for v in out[0]))
        return out

    def w_opt(**kw):
        B, max_b = orig["optimize"](**kw)
        chain = []
        b = max_b
        while b != (-1, -1, -1):
            chain.append([int(v) for v in b])
            b = B[b]
        rec["problems"].append(dict(interval=len(rec["final_c"]), start=int(kw["start"]), end=int(kw["end"]), chain=chain))
        return B, max_b

    def w_runopt(**kw):
        out = orig["run_optimize"](**kw)
        rec["final_c"].append([int(v) for v in out])
        return out

    def w_refine(y_raw, y_idxs, sigma, **kw):
        state["in_refine"] = True
        try:
            out = orig["refine_segmentation"](y_raw, y_idxs, sigma, **kw)
        finally:
            state["in_refine"] = False
        rec["refine"].append([int(v) for v in out])
        return out

    def w_segment(tint, *a, **kw):
        rec["tint"] = tint
        return orig["segment"](tint, *a, **kw)

    wrappers = dict(process_splicing_data=w_psd, gaussian_filter1d=w_gauss, candidates_from_peaks=w_cands,
                    break_large_problems=w_break, optimize=w_opt, run_optimize=w_runopt,
                    refine_segmentation=w_refine, segment=w_segment)
    for k, v in wrappers.items():
        setattr(R, k, v)
    try:
        table = R.smooth_threshold(threshold=threshold_rate)
        os.makedirs(os.path.join(outdir, contig), exist_ok=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                R.run_segment((split_dir, outdir, contig, tint_id, sigma, table, threshold_rate, variance_factor,
                               max_problem_size, min_read_support_outside, not consider_ends))
            except (AssertionError, IndexError) as e:
                if not allow_raise:
                    raise
                rec["raised"] = "%s: %s" % (type(e).__name__, e)
            vals = np.array([v for y in rec["Y"] for v in y if v > 0])
            rec["threshold"] = float(vals.mean() + variance_factor * vals.std())
        rec["h_table"] = [float(v) for v in table]
    finally:
        for k, v in orig.items():
            setattr(R, k, v)
    return rec.pop("tint"), rec
