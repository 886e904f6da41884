"""End to end on the GPU box: the drop-in CLI on split directories, output bytes vs the reference's own
segment_*.tsv (golden fixtures)."""
import os
import subprocess
import sys

import pytest

import goldens
from test_host_mirror import NAMES, input_dir

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cli(split_dir, out_dir, run):
    cmd = [sys.executable, os.path.join(ROOT, "py", "freddie_segment.py"), "-s", split_dir, "-o", out_dir, "--gpus", "1",
           "-sd", str(run["sigma"]), "-tp", str(run["threshold_rate"]), "-vf", str(run["variance_factor"]),
           "-mps", str(run["max_problem_size"]), "-lo", str(run["min_read_support_outside"])]
    if run["consider_ends"]:
        cmd.append("--consider-ends")
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    return res.stdout


@pytest.mark.parametrize("name", NAMES)
def test_cli_output_is_byte_identical_to_reference(name, tmp_path):
    g = goldens.load(name)
    case = goldens.manifest()["cases"][name]
    d, contig, tid = input_dir(name, tmp_path)
    out = str(tmp_path / "out")
    stdout = run_cli(d, out, case["run"])
    assert "[freddie_segment] Done with 0/1 tints (0.0%)" in stdout
    got = open(os.path.join(out, contig, "segment_%s_%d.tsv" % (contig, tid)), "rb").read()
    assert got == g["segment_tsv"].tobytes()
    assert os.path.exists(os.path.join(out, contig, "segment_%s_%d.log" % (contig, tid)))


def test_cli_many_partitions_in_batches(tmp_path):
    """Several partitions through the batched driver give the same files as one by one."""
    from freddie_amd import synth
    d = str(tmp_path / "in")
    for i in range(6):
        synth.generate(100 + i, n_reads=120 + 10 * i, n_exons=40, rp=0.1, write_dir=d)
    run = dict(sigma=5.0, threshold_rate=0.9, variance_factor=3.0, max_problem_size=50, min_read_support_outside=3,
               consider_ends=False)
    run_cli(d, str(tmp_path / "all"), run)
    for i in range(6):
        one = str(tmp_path / ("in%d" % i))
        synth.generate(100 + i, n_reads=120 + 10 * i, n_exons=40, rp=0.1, write_dir=one)
        run_cli(one, str(tmp_path / ("out%d" % i)), run)
        a = open(os.path.join(str(tmp_path / "all"), "chrS", "segment_chrS_%d.tsv" % (100 + i)), "rb").read()
        b = open(os.path.join(str(tmp_path / ("out%d" % i)), "chrS", "segment_chrS_%d.tsv" % (100 + i)), "rb").read()
        assert a == b and len(a) > 0


def test_in_process_seam_matches_reference(tmp_path):
    """segment(tint, ...) -- the reference's in-process seam (py/freddie_segment.py:738-747) -- through the Python
    host code: same mutations of the tint dict, same TSV bytes."""
    from freddie_amd import segment, tables
    name = "g1_retention"
    g = goldens.load(name)
    case = goldens.manifest()["cases"][name]
    d, contig, tid = input_dir(name, tmp_path)
    tint = segment._load_partition(d, contig, tid)
    run = case["run"]
    rid = segment.segment(tint, run["sigma"], tables.smooth_threshold(run["threshold_rate"]), run["threshold_rate"],
                          run["variance_factor"], run["max_problem_size"], run["min_read_support_outside"],
                          not run["consider_ends"])
    assert rid == tint["id"]
    assert tint["final_positions"] == g["final_positions"].tolist()
    assert tint["segs"][0] == (tint["final_positions"][0], tint["final_positions"][1])
    out = tmp_path / "seam.tsv"
    segment.write_segment_tsv(tint, str(out))
    assert out.read_bytes() == g["segment_tsv"].tobytes()


def test_cli_hundred_thousand_reads(tmp_path):
    """200 partitions x 500 reads through the batched, pipelined CLI; every output file must hash to what the
    reference CLI wrote for the same (regenerated) inputs (tests/golden/g5_cli_hashes.json)."""
    import hashlib
    import json
    from freddie_amd import synth
    doc = json.load(open(os.path.join(goldens.GOLDEN_DIR, "g5_cli_hashes.json")))
    gen = doc["generator"]
    d = str(tmp_path / "in")
    for i in range(gen["n_partitions"]):
        synth.generate(i, n_reads=gen["n_reads"], n_exons=gen["n_exons"], rp=gen["rp"], write_dir=d)
    for f, want in list(doc["split_inputs"].items())[:5]:
        assert hashlib.sha256(open(os.path.join(d, "chrS", f), "rb").read()).hexdigest() == want
    out = str(tmp_path / "out")
    cmd = [sys.executable, os.path.join(ROOT, "py", "freddie_segment.py"), "-s", d, "-o", out, "--gpus", "1", "-t", "4",
           "--batch-reads", "20000"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    bad = [f for f, want in doc["outputs"].items()
           if hashlib.sha256(open(os.path.join(out, "chrS", f), "rb").read()).hexdigest() != want]
    assert not bad, "%d of %d outputs differ from the reference, e.g. %s" % (len(bad), len(doc["outputs"]), bad[:3])


def test_cli_multi_worker_scatter(tmp_path):
    """The multi-GPU driver (one spawned worker per device entry, static LPT scatter, no collective), exercised on
    a one-GPU box by listing device 0 twice: outputs must equal the single-worker run."""
    from freddie_amd import synth
    d = str(tmp_path / "in")
    for i in range(10):
        synth.generate(300 + i, n_reads=100 + 20 * i, n_exons=50, rp=0.1, write_dir=d)
    base = [sys.executable, os.path.join(ROOT, "py", "freddie_segment.py"), "-s", d, "-t", "2", "--batch-reads", "300"]
    one = subprocess.run(base + ["-o", str(tmp_path / "one"), "--devices", "0"], capture_output=True, text=True)
    two = subprocess.run(base + ["-o", str(tmp_path / "two"), "--devices", "0,0"], capture_output=True, text=True)
    assert one.returncode == 0, one.stderr[-1500:]
    assert two.returncode == 0, two.stderr[-1500:]
    assert two.stdout.count("Done with") >= 1
    for i in range(10):
        f = "segment_chrS_%d.tsv" % (300 + i)
        a = open(os.path.join(str(tmp_path / "one"), "chrS", f), "rb").read()
        b = open(os.path.join(str(tmp_path / "two"), "chrS", f), "rb").read()
        assert a == b and len(a) > 0


def test_cli_sidecar_runs_are_byte_identical(tmp_path):
    """Row N2: `--sidecar write` on the first pass, `auto` on the second (now loading the side-cars, no TSV parsing),
    `off` as the control -- all three give the reference's bytes for a golden case and identical files for synthetic
    partitions."""
    from freddie_amd import synth
    name = "g1_retention"
    g = goldens.load(name)
    d, contig, tid = input_dir(name, tmp_path)
    for i in range(4):
        synth.generate(500 + i, n_reads=150, n_exons=40, rp=0.1, write_dir=d, contig=contig)
    case = goldens.manifest()["cases"][name]["run"]
    outs = {}
    for mode in ("write", "auto", "off"):
        out = str(tmp_path / mode)
        cmd = [sys.executable, os.path.join(ROOT, "py", "freddie_segment.py"), "-s", d, "-o", out, "--gpus", "1", "-t", "2",
               "--sidecar", mode, "-sd", str(case["sigma"]), "-tp", str(case["threshold_rate"]),
               "-vf", str(case["variance_factor"]), "-mps", str(case["max_problem_size"]),
               "-lo", str(case["min_read_support_outside"])]
        res = subprocess.run(cmd, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        outs[mode] = {f: open(os.path.join(out, contig, f), "rb").read() for f in sorted(os.listdir(os.path.join(out, contig)))
                      if f.endswith(".tsv")}
        if mode == "write":
            assert len([f for f in os.listdir(os.path.join(d, contig)) if f.endswith(".fsc")]) == 5
    assert outs["write"] == outs["auto"] == outs["off"] and len(outs["off"]) == 5
    assert outs["auto"]["segment_%s_%d.tsv" % (contig, tid)] == g["segment_tsv"].tobytes()
