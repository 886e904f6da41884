"""Native host I/O (libfreddie_host.so: parser + read_reps grouping, gaps/poly-A, writer) against the golden
fixtures -- no GPU: the reference's labels are fed to the native writer, output bytes vs the reference's TSV."""
import os

import numpy as np
import pytest

import goldens
from freddie_amd import _host
from test_host_mirror import NAMES, input_dir


def paths(d, contig, tid):
    return (os.path.join(d, contig, "split_%s_%d.tsv" % (contig, tid)), os.path.join(d, contig, "reads_%s_%d.tsv" % (contig, tid)))


@pytest.mark.parametrize("name", NAMES)
def test_native_parse_and_write_match_reference(name, tmp_path):
    g = goldens.load(name)
    d, contig, tid = input_dir(name, tmp_path)
    sp, rp = paths(d, contig, tid)
    hb = _host.HostBatch([sp], [rp], n_threads=2)
    try:
        a = hb.arrays()
        assert hb.n_reads == len(g["read_rep"])
        for key in ("iv_start", "iv_end", "rep_weight", "rep_exon_off", "ex_ts", "ex_te"):
            assert np.array_equal(a[key], g[key]), key
        F = len(g["final_positions"])
        out = str(tmp_path / "out.tsv")
        hb.write(np.array([0, F]), g["final_positions"], np.array([0, g["labels"].size]),
                 (g["labels"] + ord("0")).astype(np.uint8).ravel(), [out], n_threads=2)
        assert open(out, "rb").read() == g["segment_tsv"].tobytes()
    finally:
        hb.close()


def test_native_batch_of_partitions_threads(tmp_path):
    """Several partitions at once, 4 threads: arrays are the concatenation of the per-partition arrays."""
    names = [n for n in NAMES if n.startswith("g1") or n.startswith("e_")]
    sps, rps, gs = [], [], []
    for n in names:
        d, contig, tid = input_dir(n, tmp_path)
        sp, rp = paths(d, contig, tid)
        sps.append(sp); rps.append(rp); gs.append(goldens.load(n))
    hb = _host.HostBatch(sps, rps, n_threads=4)
    try:
        a = hb.arrays()
        assert np.array_equal(a["iv_start"], np.concatenate([g["iv_start"] for g in gs]))
        assert np.array_equal(a["rep_weight"], np.concatenate([g["rep_weight"] for g in gs]))
        assert np.array_equal(a["ex_ts"], np.concatenate([g["ex_ts"] for g in gs]))
        assert np.array_equal(np.diff(a["part_rep_off"]), [len(g["rep_weight"]) for g in gs])
        pfo = np.zeros(len(gs) + 1, np.int64); lo = np.zeros(len(gs) + 1, np.int64)
        np.cumsum([len(g["final_positions"]) for g in gs], out=pfo[1:])
        np.cumsum([g["labels"].size for g in gs], out=lo[1:])
        outs = [str(tmp_path / ("o%d.tsv" % i)) for i in range(len(gs))]
        hb.write(pfo, np.concatenate([g["final_positions"] for g in gs]), lo,
                 np.concatenate([(g["labels"] + 48).astype(np.uint8).ravel() for g in gs]), outs, n_threads=4)
        for o, g in zip(outs, gs):
            assert open(o, "rb").read() == g["segment_tsv"].tobytes()
    finally:
        hb.close()


def test_native_parser_rejects_malformed(tmp_path):
    sp = tmp_path / "split_c_1.tsv"; rp = tmp_path / "reads_c_1.tsv"
    rp.write_text("0\tc\t1\tACGT\n")
    for bad in ("#c\t1\t10-20\t1\n0\tr\tc\t+\t1\t10-20:0-10:10Q\n",      # bad CIGAR op
                "#c\t1\t10-20,15-30\t0\n",                                 # overlapping intervals
                "#c\t1\t10-20\t2\n0\tr\tc\t+\t1\t10-20:0-10:10M\n",       # read_count mismatch
                "#c\t1\t10-20\t1\n0\tr@x\tc\t+\t1\t10-20:0-10:10M\n"):    # '@' is not allowed in names
        sp.write_text(bad)
        with pytest.raises(_host.HostError):
            _host.HostBatch([str(sp)], [str(rp)])
