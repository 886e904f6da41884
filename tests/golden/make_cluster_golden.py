#!/usr/bin/env python3
"""Mint the fixtures of the clustering stage's pre-ILP host functions by executing the REFERENCE's own source of
read_segment(), find_segment_read(), preprocess_ilp(), split_list_evenly() and partition_reads()
(vpc-ccg/freddie py/freddie_cluster.py:112-116, :119-183, :196-274, :277-328).

The module itself cannot be imported here (it imports gurobipy at the top, :13, not installed), so only the top-level
statements those functions need are taken from its syntax tree -- the regex constants (:15-34), the functions themselves and
the two garbage-cost helpers -- and executed unmodified, with the module's own networkx imports (:11-12; networkx 3.4.2 is
installed in this image).  Nothing is stood in for anything.
Inputs: the reference's own segment_*.tsv bytes already stored in tests/golden/*.npz, and -- for partition_reads(), whose
compatibility graphs are nearly empty on those -- seeded random preprocessed tints from tests/cluster_util.random_tint
(our generator: only its seeds and shapes are stored; the stored OUTPUTS are the reference function's).
Build container only.

Usage: python tests/golden/make_cluster_golden.py
"""
import ast
import contextlib
import copy
import gzip
import io
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/py/freddie_cluster.py"
WANT_FUNCS = {"read_segment", "find_segment_read", "preprocess_ilp", "garbage_cost_introns", "garbage_cost_exons",
              "split_list_evenly", "partition_reads"}
ILP_SIZES = (7, 1000)          # maximum_ilp_size: 7 forces the even split of large components (:259), 1000 is the CLI default
# (seed, reps, segments, keyword arguments of cluster_util.random_tint, maximum_ilp_size values)
RANDOM_TINTS = [(100 + k, n, m, {}, (50,)) for k, (n, m) in enumerate(
                    [(1, 5), (2, 1), (63, 31), (64, 32), (65, 33), (130, 64), (200, 65), (257, 100), (40, 300)])] + [
                (200, 150, 20, dict(n_isoforms=2, noise=0.0, tail_p=0.0), (50, 7)),
                (201, 120, 24, dict(n_isoforms=12, noise=0.1, tail_p=0.6), (50, 7)),
                (3, 70, 40, {}, (7, 1000)), (11, 300, 60, dict(n_isoforms=4), (7, 40, 1000))]
WANT_NAMES = {"tint_prog", "internal_gap_re", "softclip_gap_re", "poly_gap_re", "read_prog", "internal_gap_prog",
              "softclip_gap_prog", "poly_gap_prog"}


def load_reference_functions():
    tree = ast.parse(open(REF).read(), REF)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in WANT_FUNCS:
            keep.append(node)
        elif isinstance(node, ast.Assign) and all(isinstance(t, ast.Name) and t.id in WANT_NAMES for t in node.targets):
            keep.append(node)
    ns = {"__name__": "freddie_cluster_partial"}
    exec("import re\nfrom math import ceil, floor\nfrom networkx.algorithms import components\nfrom networkx import Graph\n", ns)
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    assert WANT_FUNCS <= set(ns) and WANT_NAMES <= set(ns)
    return ns


def canon(tint):
    """JSON-able canonical form of a preprocessed tint."""
    d = tint["ilp_data"]
    n = len(tint["read_reps"])
    return dict(
        id=tint["id"], chr=tint["chr"], segs=[list(s) for s in tint["segs"]], read_reps=tint["read_reps"],
        I=["".join(map(str, d["I"][i])) for i in range(n)], C=["".join(map(str, d["C"][i])) for i in range(n)],
        FL=[list(d["FL"][i]) for i in range(n)], garbage_cost=[d["garbage_cost"][i] for i in range(n)],
        reads=[dict(id=r["id"], name=r["name"], chr=r["chr"], strand=r["strand"], tint=r["tint"],
                    data="".join(map(str, r["data"])),
                    gaps=sorted([list(k) + [v] for k, v in r["gaps"].items()]),
                    softclip=sorted(r["softclip"].items()), poly_tail=sorted([k, list(v)] for k, v in r["poly_tail"].items()),
                    poly_tail_category=r["poly_tail_category"]) for r in tint["reads"]])


def partitions_of(ns, tint, size):
    """tint['partitions'] as the reference's partition_reads() leaves it (its progress print, :262, swallowed)."""
    t = copy.deepcopy(tint)
    with contextlib.redirect_stdout(io.StringIO()):
        ns["partition_reads"](t, size)
    return [[list(rids), [list(pair) for pair in incomp]] for rids, incomp in t["partitions"]]


def main():
    ns = load_reference_functions()
    out_dir = os.path.join(HERE, "cluster")
    os.makedirs(out_dir, exist_ok=True)
    index = {}
    for f in sorted(os.listdir(HERE)):
        if not f.endswith(".npz"):
            continue
        g = np.load(os.path.join(HERE, f))
        tsv = g["segment_tsv"].tobytes()
        if not tsv:
            continue
        with tempfile.NamedTemporaryFile("wb", suffix=".tsv", delete=False) as tmp:
            tmp.write(tsv)
        try:
            tints = ns["read_segment"](tmp.name)
        finally:
            os.unlink(tmp.name)
        assert len(tints) == 1
        tint = list(tints.values())[0]
        ns["preprocess_ilp"](tint, dict(recycle_model="constant"))
        doc = canon(copy.deepcopy(tint))
        doc["partitions"] = {str(m): partitions_of(ns, tint, m) for m in ILP_SIZES}
        name = f[:-4]
        text = json.dumps(doc, sort_keys=True, separators=(",", ":"))
        with gzip.GzipFile(os.path.join(out_dir, name + ".json.gz"), "wb", mtime=0) as fz:
            fz.write(text.encode())
        index[name] = dict(segment_tsv_sha256=hashlib.sha256(tsv).hexdigest(), n_reads=len(doc["reads"]),
                           n_reps=len(doc["read_reps"]), n_segs=len(doc["segs"]),
                           n_tails=sum(r["poly_tail_category"] != "N" for r in doc["reads"]),
                           n_partitions={m: len(v) for m, v in doc["partitions"].items()})
        print(name, index[name])
    # partition_reads() on random preprocessed tints (graphs with real structure: several pruning passes, split components)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import cluster_util
    rnd = []
    for seed, n_reps, n_segs, kw, sizes in RANDOM_TINTS:
        tint = cluster_util.random_tint(seed, n_reps, n_segs, **kw)
        rnd.append(dict(seed=seed, n_reps=n_reps, n_segs=n_segs, kw=kw,
                        partitions={str(m): partitions_of(ns, tint, m) for m in sizes}))
        print("random tint", seed, n_reps, n_segs, {m: len(v) for m, v in rnd[-1]["partitions"].items()})
    with gzip.GzipFile(os.path.join(out_dir, "partition_random.json.gz"), "wb", mtime=0) as fz:
        fz.write(json.dumps(rnd, sort_keys=True, separators=(",", ":")).encode())
    import networkx
    json.dump(dict(reference="vpc-ccg/freddie py/freddie_cluster.py read_segment/preprocess_ilp/partition_reads (source executed from /root/reference)",
                   networkx=networkx.__version__, cases=index), open(os.path.join(out_dir, "INDEX.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
