"""Differential test of the oracle against the live reference; only where /root/reference exists
(the build container).  Skipped on the GPU box, where the reference is absent."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import refrun  # noqa: E402
import util  # noqa: E402
from freddie_amd import pack, synth  # noqa: E402

pytestmark = pytest.mark.skipif(not refrun.available(), reason="reference not present")


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_oracle_vs_live_reference(tmp_path, seed):
    rng = np.random.default_rng(seed)
    gen = dict(n_reads=int(rng.integers(50, 400)), n_exons=int(rng.integers(5, 120)), rp=float(rng.choice([0, .05, .3])),
               jp=float(rng.choice([0, .3, .8])), jsd=float(rng.choice([2, 6])), max_span=int(rng.choice([0, 6, 14])))
    run = dict(sigma=float(rng.choice([3.0, 5.0, 8.0])), threshold_rate=float(rng.choice([.8, .9, .95])),
               min_read_support_outside=int(rng.choice([0, 3, 10])), max_problem_size=int(rng.choice([20, 50])))
    synth.generate(seed, write_dir=str(tmp_path / "in"), **gen)
    tint, rec = refrun.run_recorded(str(tmp_path / "in"), str(tmp_path / "out"), "chrS", seed, **run)
    g = synth.generate(seed, with_seq=False, **gen)
    part = pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te)
    assert part.n_reps == len(tint["read_reps"])
    o = util.run_oracle(part, dict(run, ignore_ends=True))
    assert o["error"] == 0, o["errmsg"]
    assert np.array_equal(np.concatenate(rec["Y"]), o["Y"])
    assert np.array_equal(np.array(tint["final_positions"], np.int32), o["final_pos"])
    for ri, (_, ridxs) in enumerate(tint["read_reps"]):
        assert list(o["labels"][ri]) == tint["reads"][ridxs[0]]["data"]


def test_oracle_vs_live_reference_threshold_rate_one(tmp_path):
    """threshold_rate == 1 (the CLI's upper bound): long segments get h == 1, low == 0, so a read without coverage
    is ambiguous ('2') there -- the lo < 0 paths of the kernels."""
    seed = 7
    gen = dict(n_reads=300, n_exons=60, rp=0.1)
    run = dict(sigma=5.0, threshold_rate=1.0, min_read_support_outside=3, max_problem_size=50)
    synth.generate(seed, write_dir=str(tmp_path / "in"), **gen)
    tint, rec = refrun.run_recorded(str(tmp_path / "in"), str(tmp_path / "out"), "chrS", seed, **run)
    g = synth.generate(seed, with_seq=False, **gen)
    part = pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te)
    o = util.run_oracle(part, dict(run, ignore_ends=True))
    assert o["error"] == 0, o["errmsg"]
    assert np.array_equal(np.array(tint["final_positions"], np.int32), o["final_pos"])
    n2 = 0
    for ri, (_, ridxs) in enumerate(tint["read_reps"]):
        assert list(o["labels"][ri]) == tint["reads"][ridxs[0]]["data"]
        n2 += list(o["labels"][ri]).count(2)
    assert n2 > 0


def test_oracle_vs_live_reference_large_problems(tmp_path):
    """max_problem_size = 100: DP problems with up to ~110 candidates (the default 50 keeps them <= 60)."""
    seed = 5
    gen = dict(n_reads=300, n_exons=150, rp=0.3, max_span=0)
    run = dict(sigma=5.0, threshold_rate=0.9, min_read_support_outside=3, max_problem_size=100, variance_factor=9.0)
    synth.generate(seed, write_dir=str(tmp_path / "in"), **gen)
    tint, rec = refrun.run_recorded(str(tmp_path / "in"), str(tmp_path / "out"), "chrS", seed, **run)
    g = synth.generate(seed, with_seq=False, **gen)
    part = pack.pack_partition(g.iv_start, g.iv_end, g.read_exon_off, g.ex_ts, g.ex_te)
    o = util.run_oracle(part, dict(run, ignore_ends=True))
    assert o["error"] == 0, o["errmsg"]
    assert int((np.asarray(o["prob_end"]) - np.asarray(o["prob_start"]) + 1).max()) > 60
    assert np.array_equal(np.array(tint["final_positions"], np.int32), o["final_pos"])
    for ri, (_, ridxs) in enumerate(tint["read_reps"]):
        assert list(o["labels"][ri]) == tint["reads"][ridxs[0]]["data"]


@pytest.mark.parametrize("seed", [31, 33])
def test_native_writer_vs_live_reference_on_crafted_poly_clips(tmp_path, seed):
    """The native writer's gaps / poly-A annotation (:370-472; one pass for both letters since round 6) against the reference's own
    run_segment() output bytes on soft clips crafted to hold competing poly-A / poly-T runs (ties, lengths 19 / 20, impurities, both
    strands).  Labels and final positions are the reference's."""
    from freddie_amd import _host
    from test_host_native import craft_poly_clips, paths
    d = str(tmp_path / "in")
    synth.generate(seed, write_dir=d, n_reads=240, n_exons=30, rp=0.1)
    craft_poly_clips(d, "chrS", seed, seed)
    tint, rec = refrun.run_recorded(d, str(tmp_path / "out"), "chrS", seed)
    want = open(os.path.join(str(tmp_path / "out"), "chrS", "segment_chrS_%d.tsv" % seed), "rb").read()
    assert sum(want.count(k) for k in (b"SA_", b"ST_", b"EA_", b"ET_")) >= 25
    labels = np.array([tint["reads"][ridxs[0]]["data"] for _, ridxs in tint["read_reps"]], np.uint8)
    fp = np.array(tint["final_positions"], np.int32)
    sp, rp = paths(d, "chrS", seed)
    hb = _host.HostBatch([sp], [rp], n_threads=2)
    try:
        out = str(tmp_path / "got.tsv")
        hb.write(np.array([0, len(fp)]), fp, np.array([0, labels.size]), (labels + 48).astype(np.uint8).ravel(), [out], n_threads=2)
        assert open(out, "rb").read() == want
    finally:
        hb.close()
