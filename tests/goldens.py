"""Loader of the golden fixtures (outputs of the reference itself, minted by tests/golden/make_golden.py)."""
import glob
import json
import os

import numpy as np

from freddie_amd import pack

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(n for n in (os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))) if not n.startswith("x_"))


def raising_names():
    """Fixtures of inputs on which the reference itself raises (x_*.npz: inputs + the exception's text)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "x_*.npz")))


def manifest():
    return json.load(open(os.path.join(GOLDEN_DIR, "MANIFEST.json")))


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def partition_of(g):
    return pack.PackedPartition(g["iv_start"], g["iv_end"], g["rep_weight"], g["rep_exon_off"], g["ex_ts"], g["ex_te"],
                                g["read_rep"])


def params_of(g):
    return dict(sigma=float(g["sigma"]), threshold_rate=float(g["threshold_rate"]),
                variance_factor=float(g["variance_factor"]), max_problem_size=int(g["max_problem_size"]),
                min_read_support_outside=int(g["min_read_support_outside"]), ignore_ends=bool(g["ignore_ends"]))


def tables_of(g):
    return dict(w_main=g["w_main"], w_refine=g["w_refine"], h_table=g["h_table"])


def as_oracle_result(g):
    """The golden in the layout of oracle.segment()'s result, so util.compare_partitions() can take either."""
    K = len(g["iv_start"])
    pos_off = np.zeros(K + 1, np.int64)
    np.cumsum(g["iv_end"].astype(np.int64) - g["iv_start"] + 1, out=pos_off[1:])
    final_pos = g["final_positions"]
    # final positions per interval
    final_off = np.zeros(K + 1, np.int64)
    for k in range(K):
        final_off[k + 1] = final_off[k] + np.count_nonzero((final_pos >= g["iv_start"][k]) & (final_pos <= g["iv_end"][k]))
    final_y = final_pos.copy()
    for k in range(K):
        final_y[final_off[k]:final_off[k + 1]] -= g["iv_start"][k]
    return dict(error=0, errmsg="", pos_off=pos_off, Y_raw=g["Y_raw"].astype(np.float64), Y=g["Y"],
                threshold=float(g["threshold"]), cand_off=g["cand_off"], cands=g["cands"], fixed_off=g["fixed_off"],
                fixed=g["fixed"], finalc_off=g["finalc_off"], finalc=g["finalc"], refine_off=g["refine_off"],
                refine=g["refine"], final_off=final_off, final_y=final_y, final_pos=final_pos, labels=g["labels"])
