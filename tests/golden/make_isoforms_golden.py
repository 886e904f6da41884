#!/usr/bin/env python3
"""Mint the fixtures of the isoform-consensus stage (row N4) by running the REFERENCE here: py/freddie_isoforms.py
needs only the standard library and is imported from /root/reference as it is.

Inputs: cluster_*.tsv files in the format of the reference's output_isoforms() (py/freddie_cluster.py:639-693), written
here from the reference's own segment TSVs (tests/golden/*.npz) with a simple nearest-centre grouping standing for the
ILP's assignment (the ILP needs Gurobi; any assignment exercises the consensus stage), and the split TSVs the
segment goldens were minted from (regenerated from their seeds / the committed edge cases).
Outputs: for several (majority_threshold, correction_window) settings the GTF text the reference writes.
Stored per case in tests/golden/isoforms/<case>.json.gz: the cluster TSV text (an input) and the expected GTFs.
Build container only.   Usage: python tests/golden/make_isoforms_golden.py
"""
import gzip
import importlib.util
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import goldens  # noqa: E402
from freddie_amd import cluster_prep  # noqa: E402
from test_host_mirror import NAMES, input_dir  # noqa: E402

SETTINGS = [(0.5, 8), (0.7, 3), (0.5, 0), (1.0, 20)]


def reference_module():
    spec = importlib.util.spec_from_file_location("ref_freddie_isoforms", "/root/reference/py/freddie_isoforms.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cluster_tsv_text(segment_tsv_path, n_centres=6):
    """A cluster file for the tint of a segment TSV: reads join the nearest of a few centre reads (Hamming distance on
    the labels inside the centre's span) or the garbage isoform ('*')."""
    tint = list(cluster_prep.read_segment(segment_tsv_path).values())[0]
    cluster_prep.preprocess_ilp(tint, dict(recycle_model="constant"))
    reads = tint["reads"]
    M = len(tint["segs"])
    order = sorted(range(len(reads)), key=lambda r: (-sum(v == 1 for v in reads[r]["data"]), r))
    centres = order[:: max(1, len(order) // n_centres)][:n_centres]
    lines = ["#{}\t{}\t{}".format(tint["chr"], tint["id"], ",".join([str(s[0]) for s in tint["segs"]] + [str(tint["segs"][-1][1])]))]
    members = {k: [] for k in range(len(centres))}
    garbage = []
    for r, read in enumerate(reads):
        best, best_d = None, None
        for k, c in enumerate(centres):
            d = sum((a == 1) != (b == 1) for a, b in zip(read["data"], reads[c]["data"]))
            if best_d is None or d < best_d:
                best, best_d = k, d
        (members[best] if best_d <= max(2, M // 6) else garbage).append(r)
    for iid in range(len(centres)):
        lines.append("isoform_{}\t{}\t{}".format(iid, tint["id"], "".join(str(v % 2) for v in reads[centres[iid]]["data"])))
        for r in members[iid]:
            read = reads[r]
            data = "".join(map(str, read["data"]))
            cols = [str(read["id"]), read["name"], read["chr"], read["strand"], str(read["tint"]), "0", read["poly_tail_category"],
                    str(iid), data] + list(data) + ["{}:{}".format(k, v) for k, v in sorted(read["poly_tail"].items())]
            lines.append("\t".join(cols))
    for r in garbage:
        read = reads[r]
        data = "".join(map(str, read["data"]))
        lines.append("\t".join([str(read["id"]), read["name"], read["chr"], read["strand"], str(read["tint"]), "0",
                                read["poly_tail_category"], "*", data] + list(data)))
    return "\n".join(lines) + "\n"


def main():
    ref = reference_module()
    out_dir = os.path.join(HERE, "isoforms")
    os.makedirs(out_dir, exist_ok=True)
    index = {}
    with tempfile.TemporaryDirectory() as work:
        class P:                                             # what input_dir() needs of pytest's tmp_path
            def __init__(self, p): self.p = p
            def __truediv__(self, o): return os.path.join(self.p, o)
        for name in NAMES:
            g = goldens.load(name)
            d, contig, tid = input_dir(name, P(work))
            seg = os.path.join(work, "segment_%s.tsv" % name)
            open(seg, "wb").write(g["segment_tsv"].tobytes())
            text = cluster_tsv_text(seg)
            ctsv = os.path.join(work, "cluster_%s_%d.tsv" % (contig, tid))
            open(ctsv, "w").write(text)
            split_tsv = os.path.join(d, contig, "split_%s_%d.tsv" % (contig, tid))
            expect = {}
            for m, w in SETTINGS:
                sys.stdout = open(os.devnull, "w")
                try:
                    recs = ref.run_consensus([contig, tid, ctsv, split_tsv, m, w])
                finally:
                    sys.stdout = sys.__stdout__
                recs.sort()
                expect["%g,%d" % (m, w)] = "".join(r + "\n" for _, r in recs)
            doc = dict(contig=contig, tint_id=tid, cluster_tsv=text, gtf=expect)
            with gzip.GzipFile(os.path.join(out_dir, name + ".json.gz"), "wb", mtime=0) as fz:
                fz.write(json.dumps(doc, sort_keys=True).encode())
            index[name] = dict(transcripts={k: v.count("\ttranscript\t") for k, v in expect.items()},
                               reads=text.count("\n") - 1)
            print(name, index[name])
    json.dump(dict(reference="vpc-ccg/freddie py/freddie_isoforms.py run_consensus (imported from /root/reference)",
                   settings=SETTINGS, cases=index), open(os.path.join(out_dir, "INDEX.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
