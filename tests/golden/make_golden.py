#!/usr/bin/env python3
"""Mint the golden fixtures by running the REFERENCE (imported from /root/reference) here.

The reference has no tests or golden vectors of its own, so parity is pinned
on outputs of the reference itself.  This script can only run in the build container (the
reference never travels); what it writes is data:

  tests/golden/<case>.npz      inputs (flat arrays), parameter tables, every recorded intermediate,
                               the labels and the bytes of the reference's own segment_*.tsv
  tests/golden/edge/<case>/... hand-made split/reads TSVs of the edge cases (inputs, written here)
  tests/golden/MANIFEST.json   versions, generator arguments, sha256 of the generated input TSVs

Usage: python tests/golden/make_golden.py [--big]
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import refrun  # noqa: E402
from freddie_amd import synth, tables  # noqa: E402

RUN_DEFAULTS = dict(sigma=5.0, threshold_rate=0.9, variance_factor=3.0, max_problem_size=50,
                    min_read_support_outside=3, consider_ends=False)

SYNTH_CASES = {
    "g1_retention": (dict(index=11, n_reads=200, n_exons=150, rp=0.05), {}),
    "g1_dense": (dict(index=12, n_reads=200, n_exons=150, rp=0.0), {}),
    "g1_long": (dict(index=13, n_reads=200, n_exons=150, rp=0.05, max_span=0), {}),
    "g3_ont": (dict(index=14, n_reads=1000, n_exons=150, rp=0.08, jp=0.8, jsd=6.0), dict(sigma=3.0, threshold_rate=0.8)),
    "g_refine": (dict(index=15, n_reads=600, n_exons=60, rp=0.2, max_span=0), dict(min_read_support_outside=1000)),
    "g_weights_ends": (dict(index=16, n_reads=400, n_exons=80, rp=0.1, jp=0.0),
                       dict(consider_ends=True, max_problem_size=10, variance_factor=1.0)),
    "g_sigma12": (dict(index=17, n_reads=400, n_exons=80, rp=0.3, jp=0.5, jsd=4.0, max_span=0),
                  dict(sigma=12.0, min_read_support_outside=0)),
    "g_tiny": (dict(index=18, n_reads=6, n_exons=3, rp=0.0, jp=0.5, jsd=2.0), {}),
    # the CLI's parameter bounds (parse_args :104-109: 0 < sigma <= 50, 0.5 <= threshold_rate <= 1, 0 < variance_factor < 10,
    # max_problem_size > 3): sigma = 50 is radius 200 in the main filter (intervals shorter than the radius: repeated
    # reflection) and radius 50 in refine_segmentation; sigma = 0.1 is radius 0 in both (the filters are the identity);
    # threshold_rate = 0.5 makes smooth_threshold() six entries of 0.5 (h = l = 0.5: nothing is ambiguous but an exact half)
    "b_sigma50": (dict(index=21, n_reads=300, n_exons=40, rp=0.1), dict(sigma=50.0)),
    "b_sigma50_dense": (dict(index=22, n_reads=200, n_exons=60, rp=0.0), dict(sigma=50.0)),
    "b_sigma50_refine": (dict(index=29, n_reads=600, n_exons=60, rp=0.2, max_span=0), dict(sigma=50.0, min_read_support_outside=1000)),
    "b_sigma01": (dict(index=23, n_reads=300, n_exons=60, rp=0.1, jp=0.5), dict(sigma=0.1)),
    "b_sigma01_refine": (dict(index=30, n_reads=600, n_exons=60, rp=0.2, max_span=0), dict(sigma=0.1, min_read_support_outside=1000)),
    "b_tau05": (dict(index=24, n_reads=300, n_exons=60, rp=0.1), dict(threshold_rate=0.5)),
    "b_tau051": (dict(index=31, n_reads=300, n_exons=60, rp=0.1), dict(threshold_rate=0.51)),
    "b_vf001": (dict(index=25, n_reads=300, n_exons=60, rp=0.1), dict(variance_factor=0.01)),
    "b_vf999": (dict(index=26, n_reads=300, n_exons=60, rp=0.1), dict(variance_factor=9.99)),
    "b_mps5": (dict(index=28, n_reads=300, n_exons=60, rp=0.3, max_span=0), dict(max_problem_size=5)),
}

# Inputs on which the REFERENCE ITSELF raises (break_large_problems :636-643: with max_problem_size = 4 an anchor's window of
# ten candidates runs past the candidate list -> IndexError at :640).  The fixture holds the inputs and the exception's text;
# oracle and library must refuse the partition (x_*.npz: not in goldens.names()).
RAISE_CASES = {
    "x_mps4": (dict(index=27, n_reads=300, n_exons=60, rp=0.3, max_span=0), dict(max_problem_size=4)),
}


def seq_of(n, seed):
    rng = np.random.default_rng(seed)
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def edge_cases():
    """Hand-made partitions (contig chrE).  Each: (tint_id, header intervals, reads)."""
    cases = {}
    # single-exon reads only: nothing enters the histogram -> empty V -> NaN threshold (quirk 7)
    reads = []
    for i in range(4):
        reads.append((i, "+-"[i % 2], [(1000 + 3 * i, 1100 + 2 * i, 5, 105 - i, "%dM" % (100 - i))], 5 + 100 - i + 7))
    cases["e_single_exon"] = (1, [(1000, 1106)], reads)
    # one read rep (three identical reads), two short intervals (shorter than the Gaussian radius)
    reads = []
    for i in range(3):
        reads.append((i, "+", [(500, 512, 0, 12, "12M"), (530, 541, 12, 23, "11M")], 23 + 25))
    cases["e_one_rep"] = (2, [(500, 512), (530, 541)], reads)
    # plateau: two equal spikes two positions apart give a flat-topped smoothed peak; exon ends on the
    # interval end; exons touching (te == next ts) inside one interval
    reads = []
    for i in range(6):
        reads.append((i, "-", [(2000, 2060, 3, 63, "60M"), (2100, 2160, 63, 123, "60M"), (2162, 2230, 123, 191, "68M"),
                               (2300, 2400, 191, 291, "100M")], 291 + 4))
    for i in range(6, 10):
        reads.append((i, "+", [(2010, 2060, 0, 50, "50M"), (2060, 2120, 50, 110, "60M"), (2300, 2380, 110, 190, "80M")], 190 + 30))
    cases["e_plateau_touch"] = (3, [(2000, 2230), (2300, 2400)], reads)
    # refine_segmentation's find_peaks(distance=20) with an EXACT tie (quirk: the order of equal heights is numpy's argsort):
    # two families of 30 reads with a one-base gap at 3100|3102 and at 3110|3112.  Every read covers both sides of its gap,
    # so the DP cuts nowhere and the whole interval reaches refine_segmentation; there the two clusters smooth (radius 5) to
    # peaks at y = 101 and y = 111 of the same height to the last bit ((30 + 30) * w[1] each), ten positions apart.
    reads = []
    for i in range(30):
        reads.append((i, "+", [(3000, 3100, 0, 100, "100M"), (3102, 3390, 100, 388, "288M")], 388 + 12))
    for i in range(30, 60):
        reads.append((i, "-", [(3000, 3110, 0, 110, "110M"), (3112, 3390, 110, 388, "278M")], 388 + 9))
    cases["e_refine_tie"] = (4, [(3000, 3400)], reads)
    # threshold_rate = 0.9999: smooth_threshold() rounds its last entry (segment length 107) to 1.0 while the rate itself stays
    # below 1 -- for a segment of exactly 107 positions h = 1, l = 0, so nothing is ever "not covered" (c < 0 is false) and a read
    # WITHOUT coverage there is labelled '2' (:816-828); every other length has l > 0 and such a read gets '0'.  First interval:
    # 107 positions without an inner candidate (one segment of length 107); second interval: ordinary splicing.  Reads of
    # the second family never touch the first interval.
    reads = []
    for i in range(12):
        reads.append((i, "+", [(5000, 5106, 0, 106, "106M"), (5300, 5380, 106, 186, "80M"), (5420, 5600, 186, 366, "180M")], 366 + 10))
    for i in range(12, 20):
        reads.append((i, "-", [(5300, 5380, 4, 84, "80M"), (5420, 5500, 84, 164, "80M"), (5530, 5600, 164, 234, "70M")], 234 + 8))
    for i in range(20, 24):
        reads.append((i, "+", [(5040, 5106, 0, 66, "66M"), (5300, 5360, 66, 126, "60M")], 126 + 8))
    cases["e_tau9999_len107"] = (5, [(5000, 5106), (5300, 5600)], reads, dict(threshold_rate=0.9999))
    return cases


def write_edge(case_dir, contig, tint_id, intervals, reads, seed):
    d = os.path.join(case_dir, contig)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "split_%s_%d.tsv" % (contig, tint_id)), "w") as f:
        f.write("#%s\t%d\t%s\t%d\n" % (contig, tint_id, ",".join("%d-%d" % iv for iv in intervals), len(reads)))
        for rid, strand, ivs, _ in reads:
            f.write("%d\tedge_%d_%d\t%s\t%s\t%d" % (rid, tint_id, rid, contig, strand, tint_id))
            for ts, te, qs, qe, cig in ivs:
                f.write("\t%d-%d:%d-%d:%s" % (ts, te, qs, qe, cig))
            f.write("\n")
    with open(os.path.join(d, "reads_%s_%d.tsv" % (contig, tint_id)), "w") as f:
        for rid, strand, ivs, length in reads:
            s = seq_of(length, seed * 1000 + rid)
            if rid % 2 == 0:   # a clean poly-A tail on every other read
                s = s[:-25] + "A" * 25 if length > 60 else s
            f.write("%d\t%s\t%d\t%s\n" % (rid, contig, tint_id, s))


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def pack_tint(tint):
    iv_s = np.array([s for s, e in tint["intervals"]], np.int32)
    iv_e = np.array([e for s, e in tint["intervals"]], np.int32)
    W = np.array([len(r) for _, r in tint["read_reps"]], np.int32)
    off = [0]; ts = []; te = []
    for rep, _ in tint["read_reps"]:
        for a, b in rep:
            ts.append(a); te.append(b)
        off.append(len(ts))
    read_rep = np.empty(len(tint["reads"]), np.int32)
    for ri, (_, ridxs) in enumerate(tint["read_reps"]):
        for x in ridxs:
            read_rep[x] = ri
    return dict(iv_start=iv_s, iv_end=iv_e, rep_weight=W, rep_exon_off=np.array(off, np.int64),
                ex_ts=np.array(ts, np.int32), ex_te=np.array(te, np.int32), read_rep=read_rep)


def cat(lists, dtype=np.int32):
    off = np.zeros(len(lists) + 1, np.int64)
    np.cumsum([len(x) for x in lists], out=off[1:])
    data = np.concatenate([np.asarray(x, dtype) for x in lists]) if len(lists) and off[-1] else np.empty(0, dtype)
    return off, data


def record_case(name, split_dir, contig, tint_id, run_kw, out_dir, manifest, store_tsv=True):
    kw = dict(RUN_DEFAULTS, **run_kw)
    tmp_out = tempfile.mkdtemp(prefix="gold_out_")
    try:
        tint, rec = refrun.run_recorded(split_dir, tmp_out, contig, tint_id, **kw)
        with open(os.path.join(tmp_out, contig, "segment_%s_%d.tsv" % (contig, tint_id)), "rb") as f:
            tsv = f.read()
    finally:
        shutil.rmtree(tmp_out, ignore_errors=True)
    arrs = pack_tint(tint)
    cand_off, cands = cat(rec["cands"])
    fixed_off, fixed = cat(rec["fixed"])
    finalc_off, finalc = cat(rec["final_c"])
    refine_off, refine = cat(rec["refine"])
    labels = np.array([tint["reads"][ridxs[0]]["data"] for _, ridxs in tint["read_reps"]], np.uint8)
    probs = np.array([[p["interval"], p["start"], p["end"], len(p["chain"])] for p in rec["problems"]], np.int32).reshape(-1, 4)
    out = dict(arrs)
    out.update(
        sigma=kw["sigma"], threshold_rate=kw["threshold_rate"], variance_factor=kw["variance_factor"],
        max_problem_size=kw["max_problem_size"], min_read_support_outside=kw["min_read_support_outside"],
        ignore_ends=int(not kw["consider_ends"]),
        w_main=tables.gaussian_half_kernel(kw["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(kw["sigma"], 1.0),
        h_table=np.array(rec["h_table"], np.float64),
        Y_raw=np.concatenate(rec["Y_raw"]).astype(np.int32), Y=np.concatenate(rec["Y"]),
        threshold=np.float64(rec["threshold"]),
        cand_off=cand_off, cands=cands, fixed_off=fixed_off, fixed=fixed, finalc_off=finalc_off, finalc=finalc,
        refine_off=refine_off, refine=refine, problems=probs,
        final_positions=np.array(tint["final_positions"], np.int32), labels=labels,
        segment_tsv=np.frombuffer(tsv if store_tsv else b"", np.uint8),
        segment_tsv_sha256=np.array(hashlib.sha256(tsv).hexdigest()),
    )
    # the reference's own smooth_threshold must agree with the product's table builder
    assert list(rec["h_table"]) == tables.smooth_threshold(kw["threshold_rate"])
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **out)
    manifest["cases"][name] = dict(run=kw, contig=contig, tint_id=tint_id, n_reads=len(tint["reads"]),
                                   n_reps=len(tint["read_reps"]), n_final=len(tint["final_positions"]),
                                   n_refined=int(len(refine)), n_problems=int(len(probs)),
                                   segment_tsv_sha256=hashlib.sha256(tsv).hexdigest())
    print("%-18s reads=%d reps=%d K=%d N=%d F=%d refined=%d problems=%d max_n=%d" % (
        name, len(tint["reads"]), len(tint["read_reps"]), len(tint["intervals"]), len(cands), len(tint["final_positions"]),
        len(refine), len(probs), max([p[2] - p[1] + 1 for p in probs] + [0])))


def record_raise_case(name, split_dir, contig, tint_id, run_kw, out_dir, manifest):
    kw = dict(RUN_DEFAULTS, **run_kw)
    tmp_out = tempfile.mkdtemp(prefix="gold_out_")
    try:
        tint, rec = refrun.run_recorded(split_dir, tmp_out, contig, tint_id, allow_raise=True, **kw)
    finally:
        shutil.rmtree(tmp_out, ignore_errors=True)
    assert "raised" in rec, "%s: the reference did not raise" % name
    out = dict(pack_tint(tint))
    out.update(sigma=kw["sigma"], threshold_rate=kw["threshold_rate"], variance_factor=kw["variance_factor"],
               max_problem_size=kw["max_problem_size"], min_read_support_outside=kw["min_read_support_outside"],
               ignore_ends=int(not kw["consider_ends"]),
               w_main=tables.gaussian_half_kernel(kw["sigma"], 4.0), w_refine=tables.gaussian_half_kernel(kw["sigma"], 1.0),
               h_table=np.array(rec["h_table"], np.float64), raised=np.array(rec["raised"]))
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **out)
    manifest["cases"][name] = dict(run=kw, contig=contig, tint_id=tint_id, n_reads=len(tint["reads"]),
                                   n_reps=len(tint["read_reps"]), raised=rec["raised"])
    print("%-18s reads=%d reps=%d the reference raises %s" % (name, len(tint["reads"]), len(tint["read_reps"]), rec["raised"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also mint the config-2-scale hash fixture (minutes, GBs)")
    ap.add_argument("--only", default="", help="mint only the cases whose name starts with this prefix (e.g. b_, x_, e_tau) "
                    "and merge them into the manifest: the other fixtures' files stay as they are")
    args = ap.parse_args()
    manifest = dict(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0],
                    reference="vpc-ccg/freddie py/freddie_segment.py (imported from /root/reference)", cases={})
    work = tempfile.mkdtemp(prefix="gold_in_")
    try:
        for name, (gen_kw, run_kw) in list(SYNTH_CASES.items()) + list(RAISE_CASES.items()):
            if args.only and not name.startswith(args.only):
                continue
            d = os.path.join(work, name)
            gkw = dict(gen_kw); idx = gkw.pop("index")
            synth.generate(idx, write_dir=d, contig="chrS", **gkw)
            if name in RAISE_CASES:
                record_raise_case(name, d, "chrS", idx, run_kw, HERE, manifest)
            else:
                record_case(name, d, "chrS", idx, run_kw, HERE, manifest)
            manifest["cases"][name]["generator"] = gen_kw
            manifest["cases"][name]["split_sha256"] = sha256_file(os.path.join(d, "chrS", "split_chrS_%d.tsv" % idx))
            manifest["cases"][name]["reads_sha256"] = sha256_file(os.path.join(d, "chrS", "reads_chrS_%d.tsv" % idx))
        edge_root = os.path.join(HERE, "edge")
        if not args.only:
            shutil.rmtree(edge_root, ignore_errors=True)
        for name, case in edge_cases().items():
            if args.only and not name.startswith(args.only):
                continue
            tint_id, intervals, reads = case[:3]
            d = os.path.join(edge_root, name)
            shutil.rmtree(d, ignore_errors=True)
            write_edge(d, "chrE", tint_id, intervals, reads, seed=tint_id)
            record_case(name, d, "chrE", tint_id, case[3] if len(case) > 3 else {}, HERE, manifest)
        if args.big:
            d = os.path.join(work, "g4_config2")
            gkw = dict(synth.WORKLOADS["config2"]); gkw.pop("n_partitions")
            synth.generate(0, write_dir=d, contig="chrS", **gkw)
            record_case("g4_config2", d, "chrS", 0, {}, HERE, manifest, store_tsv=False)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    old = {}
    mp = os.path.join(HERE, "MANIFEST.json")
    if os.path.exists(mp) and (not args.big or args.only):
        old = json.load(open(mp)).get("cases", {})
        for k, v in old.items():
            if k == "g4_config2" or args.only:
                manifest["cases"].setdefault(k, v)
    json.dump(manifest, open(mp, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
