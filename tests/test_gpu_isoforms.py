"""Isoform-consensus stage (row N4) on the GPU, through include/freddie_isoforms.h: GTF text against what the
reference itself wrote (fixtures), the CLI on a directory of tints, and the device counts against the CPU oracle on
random stage inputs that are larger than the fixtures."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest

import isoforms_util as iu
from freddie_amd import isoforms
from oracle import isoforms_oracle
from test_host_mirror import input_dir

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = isoforms.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", iu.names())
def test_gtf_matches_reference(ctx, name, tmp_path):
    doc, ctsv, split_tsv, _, _ = iu.write_case(name, tmp_path, input_dir)
    for m, w in iu.settings():
        recs = isoforms.run_consensus_batch([[doc["contig"], doc["tint_id"], ctsv, split_tsv, m, w]], ctx, verbose=False)
        recs.sort()
        assert "".join(r + "\n" for _, r in recs) == doc["gtf"]["%g,%d" % (m, w)], (name, m, w)


def test_cli_on_a_directory_of_tints(tmp_path):
    """Several tints, two contigs' worth of files in one cluster dir: the CLI's GTF = the sorted union of the reference's
    per-tint records."""
    names = ["g1_retention", "g_refine", "g_sigma12", "e_plateau_touch"]
    cdir = str(tmp_path / "cluster"); sdir = str(tmp_path / "split")
    want = []
    for name in names:
        doc = iu.load(name)
        d, contig, tid = input_dir(name, tmp_path)
        os.makedirs(os.path.join(cdir, contig), exist_ok=True); os.makedirs(os.path.join(sdir, contig), exist_ok=True)
        open(os.path.join(cdir, contig, "cluster_%s_%d.tsv" % (contig, tid)), "w").write(doc["cluster_tsv"])
        src = os.path.join(d, contig, "split_%s_%d.tsv" % (contig, tid))
        open(os.path.join(sdir, contig, "split_%s_%d.tsv" % (contig, tid)), "wb").write(open(src, "rb").read())
        want.append(doc["gtf"]["0.7,3"])
    out = str(tmp_path / "out.gtf")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "py", "freddie_isoforms.py"), "-s", sdir, "-c", cdir, "-m", "0.7",
                          "-w", "3", "-o", out, "--batch-tints", "3"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    assert res.stdout.count("Building isoforms for contig") == 4
    # the reference sorts (key, text) records over all tints: rebuild that order from the per-tint texts
    recs = []
    for text in want:
        cur = None
        for line in text.splitlines():
            f = line.split("\t")
            if f[2] == "transcript":
                if cur:
                    recs.append(cur)
                cur = [(f[0], int(f[3]) - 1), line]
            else:
                cur[1] += "\n" + line
        if cur:
            recs.append(cur)
    recs = sorted((k, t) for k, t in recs)
    assert open(out).read() == "".join(t + "\n" for _, t in recs)


def test_random_stage_inputs_match_the_oracle(ctx):
    jobs = [iu.random_job(1, 40, 60, 50), iu.random_job(2, 5, 700, 300), iu.random_job(3, 300, 4, 9), iu.random_job(4, 2, 3, 1)]
    want = copy.deepcopy(jobs)
    for m, w in ((0.5, 8), (0.6, 20), (1.0, 1)):
        a = copy.deepcopy(jobs); b = copy.deepcopy(want)
        isoforms.isoforms_cons_batch(a, ctx)
        for side in ("starts", "ends"):
            isoforms.correct_boundaries_batch(side, a, m, w, ctx)
        for isos, segments, reads in b:
            isoforms_oracle.isoforms_cons(isos, segments, reads)
            isoforms_oracle.correct_boundaries("starts", isos, reads, m, w)
            isoforms_oracle.correct_boundaries("ends", isos, reads, m, w)
        for (ia, _, _), (ib, _, _) in zip(a, b):
            assert ia == ib
        assert sum("starts" in i for ia, _, _ in a for i in ia.values()) > 100


def test_packed_labels_give_the_same_counts(ctx):
    """fiso_consensus_packed (labels at two bits each, rows of any length and at any bit offset) == fiso_consensus."""
    rng = np.random.default_rng(9)
    n_seg = rng.integers(1, 300, 37)
    per = rng.integers(0, 40, 37)
    iro = np.concatenate([[0], np.cumsum(per)])
    R = int(iro[-1])
    rows = [rng.choice(np.frombuffer(b"0012", np.uint8), size=int(n_seg[i]), p=[0.3, 0.3, 0.3, 0.1]) for i in range(37) for _ in range(int(per[i]))]
    lab = np.concatenate(rows)
    off = np.concatenate([[0], np.cumsum([len(r) for r in rows])])[:-1]
    tail = rng.integers(0, 3, R).astype(np.uint8)
    a = ctx.consensus(iro, n_seg, off, lab, tail)
    b = ctx.consensus(iro, n_seg, off, isoforms.pack_labels(lab), tail, packed=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert a[1].sum() > 0


def test_counts_are_plain_sums(ctx):
    """Property at a size the Python oracle is too slow for: 200 k reads; cov >= cons, per-isoform tails add up to the reads
    that cover something, votes add up to the (read boundary, isoform boundary) pairs inside the window (numpy check)."""
    rng = np.random.default_rng(5)
    n_iso, per, M = 400, 500, 120
    R = n_iso * per
    lab = rng.choice(np.frombuffer(b"0012", np.uint8), size=(R, M), p=[0.3, 0.3, 0.3, 0.1])
    lab[rng.random(R) < 0.01] = ord("0")
    tail = rng.integers(0, 3, R).astype(np.uint8)
    off, cons, cov, tails = ctx.consensus(np.arange(n_iso + 1) * per, np.full(n_iso, M), np.arange(R) * M, lab.reshape(-1), tail)
    one = lab == ord("1")
    has = one.any(1)
    first = np.where(has, one.argmax(1), M); last = np.where(has, M - 1 - one[:, ::-1].argmax(1), -1)
    first[has & (tail == 1)] = 0; last[has & (tail == 1)] = M - 1
    j = np.arange(M)[None, :]
    inside = (j >= first[:, None]) & (j <= last[:, None])
    assert np.array_equal(cov.reshape(n_iso, M), inside.reshape(n_iso, per, M).sum(1))
    assert np.array_equal(cons.reshape(n_iso, M), (inside & one).reshape(n_iso, per, M).sum(1))
    for k in range(3):
        assert np.array_equal(tails[:, k], (has & (tail == k)).reshape(n_iso, per).sum(1))
    nb, w = 30, 7
    iso_b = np.sort(rng.integers(0, 5000, (n_iso, nb)), axis=1).astype(np.int32)
    rb = rng.integers(0, 5000, (R, 4)).astype(np.int32)
    votes = ctx.boundary_votes(np.arange(n_iso + 1) * per, np.arange(n_iso + 1) * nb, iso_b.reshape(-1), np.arange(R + 1) * 4,
                               rb.reshape(-1), w)
    i = 17
    d = rb[i * per:(i + 1) * per].reshape(-1)[None, :] - iso_b[i][:, None]
    want = np.stack([(d == x).sum(1) for x in range(-w, w + 1)], axis=1)
    assert np.array_equal(votes[i * nb:(i + 1) * nb], want)


def _plain_counts(iro, n_seg, off, lab, tail):
    """cons / cov / tails of isoforms_cons (:203-232) with numpy, an isoform at a time."""
    cons, cov, tails = [], [], []
    for i in range(len(n_seg)):
        M, r0, r1 = int(n_seg[i]), int(iro[i]), int(iro[i + 1])
        rows = np.stack([lab[int(off[r]):int(off[r]) + M] for r in range(r0, r1)]) if r1 > r0 and M else np.zeros((r1 - r0, M), np.uint8)
        one = rows == ord("1")
        has = one.any(1) if M else np.zeros(r1 - r0, bool)
        first = np.where(has, one.argmax(1), M) if M else np.zeros(r1 - r0, int)
        last = np.where(has, M - 1 - one[:, ::-1].argmax(1), -1) if M else np.zeros(r1 - r0, int) - 1
        tl = tail[r0:r1]
        first = np.where(has & (tl == 1), 0, first); last = np.where(has & (tl == 1), M - 1, last)
        j = np.arange(M)[None, :]
        inside = (j >= first[:, None]) & (j <= last[:, None])
        cons.append((inside & one).sum(0)); cov.append(inside.sum(0))
        tails.append([int((has & (tl == k)).sum()) for k in range(3)])
    return np.concatenate(cons).astype(np.int32), np.concatenate(cov).astype(np.int32), np.asarray(tails, np.int32)


@pytest.mark.parametrize("rows_switch", ["1", "0"])
def test_row_lengths_and_read_counts_of_every_shape(ctx, rows_switch, monkeypatch):
    """The one-pass kernel (rows of at most 1 024 labels: a read's row is one load instruction of a group of lanes) and the
    two-pass one (FISO_ROWS=0; any row length) against plain numpy: row lengths around the 16-label lane and the group sizes,
    the longest row, isoforms without reads, reads without a '1', enough reads to flush the byte counters mid-way, raw and
    two-bit labels; a call with a longer row than 1 024 goes to the two-pass kernel whole."""
    monkeypatch.setenv("FISO_ROWS", rows_switch)
    rng = np.random.default_rng(21)
    shapes = [(1, 70), (15, 33), (16, 200), (17, 64), (31, 5), (32, 129), (33, 1), (150, 500), (160, 77), (161, 300), (255, 40), (256, 9),
              (511, 21), (513, 30), (1000, 1100), (1024, 300), (20, 33000), (48, 0), (100, 3), (0, 4)]
    for longer in (False, True):
        sh = shapes + ([(1025, 7), (3000, 12)] if longer else [])
        n_seg = np.asarray([m for m, _ in sh]); per = np.asarray([n for _, n in sh])
        iro = np.concatenate([[0], np.cumsum(per)])
        R = int(iro[-1])
        rows = []
        for m, n in sh:
            blk = rng.choice(np.frombuffer(b"0012", np.uint8), size=(n, m), p=[0.45, 0.25, 0.25, 0.05])
            if m:
                blk[rng.random(n) < 0.1] = ord("0")                       # reads without a '1'
                blk[rng.random(n) < 0.1, : m // 2] = ord("2")             # spans that start late
            rows.append(blk.reshape(-1))
        lab = np.concatenate(rows)
        off = np.concatenate([[0], np.cumsum(np.repeat(n_seg, per))])[:-1]
        tail = rng.integers(0, 3, R).astype(np.uint8)
        want = _plain_counts(iro, n_seg, off, lab, tail)
        for packed in (False, True):
            _, cons, cov, tails = ctx.consensus(iro, n_seg, off, isoforms.pack_labels(lab) if packed else lab, tail, packed=packed)
            assert np.array_equal(cons, want[0]), (longer, packed)
            assert np.array_equal(cov, want[1]), (longer, packed)
            assert np.array_equal(tails, want[2]), (longer, packed)
        assert want[0].sum() > 0
