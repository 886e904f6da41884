"""Host logic of the product (parser, rep grouping, gaps/polyA strings, TSV writer, scatter, CLI arguments)
against the golden fixtures -- no GPU: the labels of the reference are fed to the host code."""
import hashlib
import os
import shutil

import numpy as np
import pytest

import goldens
from freddie_amd import scatter, segment, synth, tables

EDGE = os.path.join(goldens.GOLDEN_DIR, "edge")


def input_dir(name, tmp_path):
    """Regenerates (synthetic) or locates (edge) the split directory of a golden case."""
    case = goldens.manifest()["cases"][name]
    if name.startswith("e_"):
        d = str(tmp_path / name)                     # a copy: side-car tests write next to the TSVs, never into the tree
        if not os.path.isdir(d):
            shutil.copytree(os.path.join(EDGE, name), d)
        return d, case["contig"], case["tint_id"]
    gen = dict(case["generator"])
    idx = gen.pop("index")
    d = str(tmp_path / name)
    synth.generate(idx, write_dir=d, contig=case["contig"], **gen)
    for kind in ("split", "reads"):
        path = os.path.join(d, case["contig"], "%s_%s_%d.tsv" % (kind, case["contig"], idx))
        assert hashlib.sha256(open(path, "rb").read()).hexdigest() == case["%s_sha256" % kind], \
            "generator no longer reproduces the input the golden was minted from"
    return d, case["contig"], idx


NAMES = [n for n in goldens.names() if n != "g4_config2"]


@pytest.mark.parametrize("name", NAMES)
def test_parse_group_gaps_and_write(name, tmp_path):
    g = goldens.load(name)
    d, contig, tid = input_dir(name, tmp_path)
    tint = segment._load_partition(d, contig, tid)
    part = segment.pack_tint(tint)
    # rep grouping in first-occurrence order, as the reference builds it
    for key in ("iv_start", "iv_end", "rep_weight", "rep_exon_off", "ex_ts", "ex_te", "read_rep"):
        assert np.array_equal(getattr(part, key), g[key]), key
    # feed the reference's labels / final positions, run the host-side string work, compare the TSV bytes
    tint["final_positions"] = g["final_positions"].tolist()
    tint["segs"] = list(zip(tint["final_positions"][:-1], tint["final_positions"][1:]))
    for ri, (_, ridxs) in enumerate(tint["read_reps"]):
        for ridx in ridxs:
            tint["reads"][ridx]["data"] = g["labels"][ri].tolist()
    for read in tint["reads"]:
        segment.unaligned_gaps_and_polyA(read, tint["segs"])
    out = tmp_path / "out.tsv"
    segment.write_segment_tsv(tint, str(out))
    assert out.read_bytes() == g["segment_tsv"].tobytes()


def test_cli_arguments_match_reference_defaults():
    a = segment.parse_args(["-s", "x"])
    assert (a.outdir, a.threads, a.sigma, a.threshold_rate, a.variance_factor, a.max_problem_size,
            a.min_read_support_outside, a.consider_ends) == ("freddie_segment/", 1, 5.0, 0.9, 3.0, 50, 3, False)
    a = segment.parse_args(["-s", "x", "--consider-ends", "-sd", "3", "-tp", "0.8", "-vf", "2", "-mps", "20", "-lo", "0",
                            "-o", "o", "-t", "4"])
    assert a.consider_ends is True and a.sigma == 3.0 and a.threshold_rate == 0.8 and a.max_problem_size == 20
    assert segment.parse_args(["-s", "x", "--consider-ends", "no"]).consider_ends is False
    for bad in (["-tp", "0.4"], ["-vf", "10"], ["-sd", "51"], ["-mps", "3"], ["-lo", "-1"], ["-t", "0"]):
        with pytest.raises(AssertionError):
            segment.parse_args(["-s", "x"] + bad)
    # what the library cannot run is refused at the door (the reference has no upper bound on -mps; the DP kernels stop at 1 024
    # candidates per problem), and abbreviated options are not guessed (py/freddie_segment.py reads --gpus / --devices itself)
    assert segment.parse_args(["-s", "x", "-mps", "1000"]).max_problem_size == 1000
    for bad in (["-mps", "1001"], ["--dev", "0,1"], ["--gpu", "2"]):
        with pytest.raises(SystemExit):
            segment.parse_args(["-s", "x"] + bad)


def test_parser_rejects_malformed_lines(tmp_path):
    p = tmp_path / "split_c_1.tsv"
    p.write_text("#c\t1\t10-20\t1\n0\tr\tc\t+\t1\t10-20:0-10:10Q\n")
    with pytest.raises(ValueError):
        segment.read_split(str(p))
    p.write_text("#c\t1\t10-20,15-30\t0\n")
    with pytest.raises(AssertionError):
        segment.read_split(str(p))


def test_lpt_scatter_is_a_partition_and_balanced():
    rng = np.random.default_rng(0)
    costs = rng.integers(1, 1000, 4000).tolist()
    for n in (1, 2, 4, 8):
        bins = scatter.lpt_scatter(costs, n)
        flat = sorted(i for b in bins for i in b)
        assert flat == list(range(len(costs)))
        loads = [sum(costs[i] for i in b) for b in bins]
        assert max(loads) - min(loads) <= max(costs)
        assert bins == scatter.lpt_scatter(costs, n)            # deterministic
        for r in range(n):
            assert scatter.rank_share(costs, r, n) == bins[r]


def test_smooth_threshold_table_is_the_reference_one():
    for name in NAMES[:3]:
        g = goldens.load(name)
        assert tables.smooth_threshold(float(g["threshold_rate"])) == g["h_table"].tolist()
