#!/usr/bin/env python3
"""Mint the fixtures of the clustering stage's pre-ILP host functions by executing the REFERENCE's own source of
read_segment(), find_segment_read() and preprocess_ilp() (vpc-ccg/freddie py/freddie_cluster.py:119-183, :277-328).

The module itself cannot be imported here (it imports networkx and gurobipy at the top, :11-13, neither installed),
so only the top-level statements those three functions need are taken from its syntax tree -- the regex constants
(:15-34), the functions themselves and the two garbage-cost helpers -- and executed unmodified.  Nothing is stood in
for the missing libraries; partition_reads() (which needs networkx) is therefore NOT covered by these fixtures.
Inputs are the reference's own segment_*.tsv bytes already stored in tests/golden/*.npz.  Build container only.

Usage: python tests/golden/make_cluster_golden.py
"""
import ast
import copy
import gzip
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/py/freddie_cluster.py"
WANT_FUNCS = {"read_segment", "find_segment_read", "preprocess_ilp", "garbage_cost_introns", "garbage_cost_exons"}
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
    exec("import re\nfrom math import ceil, floor\n", ns)
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
        name = f[:-4]
        text = json.dumps(doc, sort_keys=True, separators=(",", ":"))
        with gzip.GzipFile(os.path.join(out_dir, name + ".json.gz"), "wb", mtime=0) as fz:
            fz.write(text.encode())
        index[name] = dict(segment_tsv_sha256=hashlib.sha256(tsv).hexdigest(), n_reads=len(doc["reads"]),
                           n_reps=len(doc["read_reps"]), n_segs=len(doc["segs"]),
                           n_tails=sum(r["poly_tail_category"] != "N" for r in doc["reads"]))
        print(name, index[name])
    json.dump(dict(reference="vpc-ccg/freddie py/freddie_cluster.py read_segment/preprocess_ilp (source executed from /root/reference)",
                   cases=index), open(os.path.join(out_dir, "INDEX.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
