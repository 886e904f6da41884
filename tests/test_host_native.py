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


def test_native_writer_takes_packed_labels(tmp_path):
    """Labels at two bits each (what crosses PCIe) give the same TSV bytes as the ASCII form, for rows that start at
    every alignment inside a packed byte."""
    import util
    names = [n for n in NAMES if n.startswith("g1") or n.startswith("e_") or n.startswith("g_")]
    sps, rps, gs = [], [], []
    for n in names:
        d, contig, tid = input_dir(n, tmp_path)
        sp, rp = paths(d, contig, tid)
        sps.append(sp); rps.append(rp); gs.append(goldens.load(n))
    hb = _host.HostBatch(sps, rps, n_threads=3)
    try:
        pfo = np.zeros(len(gs) + 1, np.int64); lo = np.zeros(len(gs) + 1, np.int64)
        np.cumsum([len(g["final_positions"]) for g in gs], out=pfo[1:])
        np.cumsum([g["labels"].size for g in gs], out=lo[1:])
        lab = np.concatenate([(g["labels"] + 48).astype(np.uint8).ravel() for g in gs])
        assert len({int(x) % 4 for x in lo[:-1]}) > 1            # partitions start at different alignments
        outs = [str(tmp_path / ("p%d.tsv" % i)) for i in range(len(gs))]
        hb.write(pfo, np.concatenate([g["final_positions"] for g in gs]), lo, util.pack_labels(lab), outs, n_threads=3, packed=True)
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


# ---- binary side-car (row N2): same arrays, same output bytes, never trusted when stale or damaged ----------------
def _write_with(hb, g, out):
    F = len(g["final_positions"])
    hb.write(np.array([0, F]), g["final_positions"], np.array([0, g["labels"].size]),
             (g["labels"] + ord("0")).astype(np.uint8).ravel(), [out], n_threads=2)
    return open(out, "rb").read()


@pytest.mark.parametrize("name", NAMES)
def test_sidecar_round_trip_matches_reference(name, tmp_path):
    g = goldens.load(name)
    d, contig, tid = input_dir(name, tmp_path)
    sp, rp = paths(d, contig, tid)
    sc = sp[:-4] + ".fsc"
    hb = _host.HostBatch([sp], [rp])
    try:
        assert hb.n_from_sidecar == 0
        hb.write_sidecars([sc])
        a0 = {k: v.copy() for k, v in hb.arrays().items()}
    finally:
        hb.close()
    hb = _host.HostBatch([sp], [rp], sidecar_paths=[sc])
    try:
        assert hb.n_from_sidecar == 1 and hb.n_reads == len(g["read_rep"])
        a1 = hb.arrays()
        for key in a0:
            assert np.array_equal(a0[key], a1[key]), key
        assert _write_with(hb, g, str(tmp_path / "out.tsv")) == g["segment_tsv"].tobytes()
    finally:
        hb.close()


def test_sidecar_stale_or_damaged_falls_back_to_the_tsvs(tmp_path):
    name = NAMES[0]
    g = goldens.load(name)
    d, contig, tid = input_dir(name, tmp_path)
    sp, rp = paths(d, contig, tid)
    sc = sp[:-4] + ".fsc"
    hb = _host.HostBatch([sp], [rp]); hb.write_sidecars([sc]); hb.close()
    good = open(sc, "rb").read()
    # damaged payload: the checksum catches it
    bad = bytearray(good); bad[len(bad) // 2] ^= 0x40
    open(sc, "wb").write(bytes(bad))
    hb = _host.HostBatch([sp], [rp], sidecar_paths=[sc])
    assert hb.n_from_sidecar == 0
    assert _write_with(hb, g, str(tmp_path / "o1.tsv")) == g["segment_tsv"].tobytes()
    hb.close()
    # truncated file and foreign file
    for blob in (good[:len(good) - 8], b"not a side-car", b""):
        open(sc, "wb").write(blob)
        hb = _host.HostBatch([sp], [rp], sidecar_paths=[sc]); assert hb.n_from_sidecar == 0; hb.close()
    # intact side-car, but the TSV changed after it was written (mtime differs)
    open(sc, "wb").write(good)
    hb = _host.HostBatch([sp], [rp], sidecar_paths=[sc]); assert hb.n_from_sidecar == 1; hb.close()
    st = os.stat(rp)
    os.utime(rp, ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000))
    hb = _host.HostBatch([sp], [rp], sidecar_paths=[sc]); assert hb.n_from_sidecar == 0; hb.close()
    # missing side-car and None entries
    hb = _host.HostBatch([sp, sp], [rp, rp], sidecar_paths=[str(tmp_path / "absent.fsc"), None])
    assert hb.n_from_sidecar == 0 and hb.n_part == 2
    hb.close()


def test_sidecar_keeps_non_acgt_bytes(tmp_path):
    """Sequences with N, lower case and IUPAC letters: the exception list makes the round trip exact, so the poly-A
    search (which compares bytes) sees the same sequence either way."""
    sp = tmp_path / "split_c_7.tsv"; rp = tmp_path / "reads_c_7.tsv"
    seqs = ["ACGTNNacgtRYKM" + "A" * 30 + "N" + "A" * 5 + "CCCC" * 30 + "t" * 25,
            "T" * 26 + "G" * 140 + "N",
            "n" + "ACGT" * 40]
    lines = ["#c\t7\t100-300\t3\n"]
    for i, s in enumerate(seqs):
        lines.append("%d\tr%d\tc\t%s\t7\t120-180:40-100:60M\t200-260:100-160:60M\n" % (i, i, "+-"[i % 2]))
    sp.write_text("".join(lines))
    rp.write_text("".join("%d\tc\t7\t%s\n" % (i, s) for i, s in enumerate(seqs)))
    sc = str(tmp_path / "split_c_7.fsc")
    fp = np.array([100, 150, 190, 230, 300], np.int32)
    labels = np.frombuffer(b"1011", np.uint8)                  # one rep (all reads share their intervals)
    outs = []
    for use_sc in (False, True):
        hb = _host.HostBatch([str(sp)], [str(rp)], sidecar_paths=[sc] if use_sc else None)
        assert hb.n_from_sidecar == (1 if use_sc else 0)
        if not use_sc:
            hb.write_sidecars([sc])
        out = str(tmp_path / ("o%d.tsv" % use_sc))
        hb.write(np.array([0, 5]), fp, np.array([0, 4]), labels, [out])
        hb.close()
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and b"SSC" in outs[0]


def test_sidecar_tool_covers_a_split_directory(tmp_path):
    import subprocess
    import sys
    from freddie_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = str(tmp_path / "split")
    for i in range(5):
        synth.generate(40 + i, n_reads=60, n_exons=30, rp=0.1, write_dir=d, contig="chr%d" % (i % 2))
    tool = [sys.executable, os.path.join(root, "py", "freddie_sidecar.py"), "-s", d, "-t", "2", "--chunk", "2"]
    res = subprocess.run(tool, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-1500:]
    assert "5 partitions" in res.stdout
    fsc = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".fsc"))
    assert len(fsc) == 5 and not [f for dp, _, fs in os.walk(d) for f in fs if f.endswith(".tmp")]
    stamps = [os.stat(f).st_mtime_ns for f in fsc]
    assert subprocess.run(tool, capture_output=True, text=True).returncode == 0       # second run: nothing to redo
    assert stamps == [os.stat(f).st_mtime_ns for f in fsc]
    sp = [f[:-4] + ".tsv" for f in fsc]
    rp = [os.path.join(os.path.dirname(f), "reads_" + os.path.basename(f)[6:-4] + ".tsv") for f in fsc]
    a = _host.HostBatch(sp, rp, n_threads=2); b = _host.HostBatch(sp, rp, n_threads=2, sidecar_paths=fsc)
    assert b.n_from_sidecar == 5
    for k, v in a.arrays().items():
        assert np.array_equal(v, b.arrays()[k]), k
    a.close(); b.close()


def craft_poly_clips(src_dir, contig, tid, seed):
    """Rewrites the soft clips of a synthetic partition's reads with poly-A / poly-T runs that compete: both letters in one clip, equal
    purities (the 'A' run must win a tie), runs of 19 and of exactly 20, impure runs around the 0.85 limit, runs cut by a mismatch
    burst, clips shorter than 20 -- on both strands (a '-' read is scanned backwards for the complement letter)."""
    sp, rp = paths(src_dir, contig, tid)
    rng = np.random.default_rng(seed)
    clips = {}
    for line in open(sp):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        ivs = [x for x in f[5:] if x]
        q0 = int(ivs[0].split(":")[1].split("-")[0]); q1 = int(ivs[-1].split(":")[1].split("-")[1])
        clips[f[0]] = (q0, q1)
    blocks = ["A" * 25, "T" * 25, "A" * 19, "T" * 20, "A" * 20, "A" * 17 + "C" + "A" * 6, "T" * 10 + "G" + "T" * 12 + "C" + "T" * 9,
              "A" * 22 + "CC" + "A" * 30, "T" * 30 + "G" + "A" * 30, "A" * 20 + "G" + "T" * 20, "AAAC" * 8, "T" * 6 + "CCCC" + "T" * 24]

    def fill(n):
        out = ""
        while len(out) < n:
            out += blocks[int(rng.integers(len(blocks)))] if rng.random() < 0.8 else "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(1, 9))))
        return out[:n]
    lines = []
    for line in open(rp):
        f = line.rstrip("\n").split("\t")
        q0, q1 = clips[f[0]]
        seq = f[3]
        seq = fill(q0) + seq[q0:q1] + fill(len(seq) - q1)
        lines.append("\t".join(f[:3] + [seq]) + "\n")
    open(rp, "w").writelines(lines)


@pytest.mark.parametrize("seed", [31, 32])
def test_native_poly_tails_match_the_python_mirror(seed, tmp_path):
    """The writer's poly-A / poly-T choice (one pass for both letters, :352-367 + :397-408 / :427-439) on crafted soft clips against the
    Python mirror of the reference (itself pinned by the goldens' TSV bytes); tests/test_live_reference.py runs the same input through
    the reference where it is present."""
    from freddie_amd import segment, synth
    import util
    d = str(tmp_path / "in")
    synth.generate(seed, write_dir=d, n_reads=240, n_exons=30, rp=0.1)
    craft_poly_clips(d, "chrS", seed, seed)
    tint = segment._load_partition(d, "chrS", seed)
    part = segment.pack_tint(tint)
    o = util.run_oracle(part)
    assert o["error"] == 0
    tint["final_positions"] = o["final_pos"].tolist()
    tint["segs"] = list(zip(tint["final_positions"][:-1], tint["final_positions"][1:]))
    for ri, (_, ridxs) in enumerate(tint["read_reps"]):
        for ridx in ridxs:
            tint["reads"][ridx]["data"] = o["labels"][ri].tolist()
    for read in tint["reads"]:
        segment.unaligned_gaps_and_polyA(read, tint["segs"])
    want = tmp_path / "want.tsv"
    segment.write_segment_tsv(tint, str(want))
    text = want.read_text()
    counts = {k: text.count(k) for k in ("SA_", "ST_", "EA_", "ET_")}
    assert min(counts.values()) >= 3 and sum(counts.values()) >= 25, counts   # the crafted clips do produce poly tokens of both letters at both ends
    sp, rp = paths(d, "chrS", seed)
    hb = _host.HostBatch([sp], [rp], n_threads=1)
    try:
        F = len(o["final_pos"])
        out = str(tmp_path / "got.tsv")
        hb.write(np.array([0, F]), o["final_pos"], np.array([0, o["labels"].size]), (o["labels"] + 48).astype(np.uint8).ravel(), [out])
        assert open(out, "rb").read() == want.read_bytes()
    finally:
        hb.close()


def test_native_listing_and_touch(tmp_path):
    """fhost_discover / fhost_touch (include/freddie_host.h): the split directory's listing as the reference's main() builds it
    (:852-857: one directory per contig, split_<contig>_<tint>.tsv per partition, the tint id = the text behind the last '_') with the
    files' sizes, and the empty .log files of a batch (:695)."""
    split = tmp_path / "split"
    want = set()
    for contig, ids in (("chr1", [0, 7, 12345]), ("chr_2_x", [3]), ("empty", [])):
        (split / contig).mkdir(parents=True)
        for t in ids:
            body = b"x" * (10 + 3 * t % 97)
            (split / contig / ("split_%s_%d.tsv" % (contig, t))).write_bytes(body)
            (split / contig / ("reads_%s_%d.tsv" % (contig, t))).write_bytes(b"r")          # not a split file
            (split / contig / ("split_%s_%d.fsc" % (contig, t))).write_bytes(b"s")          # a side-car: not listed
            want.add((contig, t, len(body)))
    (split / "stray_file.tsv").write_bytes(b"")                                            # a file beside the contig directories
    contigs, found = _host.discover(str(split))
    assert sorted(contigs) == ["chr1", "chr_2_x", "empty"]
    assert set(found) == want and len(found) == len(want)
    (split / "chr1" / "split_chr1_abc.tsv").write_bytes(b"")                                # int("abc") raises in the reference's main()
    with pytest.raises(_host.HostError, match="not a number"):
        _host.discover(str(split))
    with pytest.raises(_host.HostError):
        _host.discover(str(tmp_path / "absent"))
    logs = [str(tmp_path / ("l%d.log" % i)) for i in range(40)]
    open(logs[3], "w").write("old")
    _host.touch(logs, n_threads=4)
    assert all(os.path.getsize(p) == 0 for p in logs)
    with pytest.raises(_host.HostError):
        _host.touch([str(tmp_path / "no_such_dir" / "x.log")])
