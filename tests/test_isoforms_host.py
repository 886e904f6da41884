"""Isoform-consensus stage (row N4), CPU side: the host mirror's readers and GTF writer plus the oracle's loops must
reproduce the GTF the reference itself wrote for the fixture inputs (tests/golden/make_isoforms_golden.py); the C-ABI
library's symbols; loud failure without a GPU."""
import os
import re

import pytest

import isoforms_util as iu
from freddie_amd import build, isoforms
from oracle import isoforms_oracle
from test_host_mirror import input_dir


def oracle_gtf(ctsv, split_tsv, m, w):
    segments, reads, isos = isoforms.read_cluster(ctsv)
    isoforms_oracle.isoforms_cons(isos, segments, reads)
    isoforms.read_split(split_tsv, reads)
    isoforms_oracle.correct_boundaries("starts", isos, reads, m, w)
    isoforms_oracle.correct_boundaries("ends", isos, reads, m, w)
    recs = isoforms.get_gtf_records(isos)
    recs.sort()
    return "".join(r + "\n" for _, r in recs)


@pytest.mark.parametrize("name", iu.names())
def test_oracle_and_host_mirror_reproduce_the_reference_gtf(name, tmp_path):
    doc, ctsv, split_tsv, _, _ = iu.write_case(name, tmp_path, input_dir)
    for m, w in iu.settings():
        assert oracle_gtf(ctsv, split_tsv, m, w) == doc["gtf"]["%g,%d" % (m, w)], (name, m, w)


def test_fixtures_exercise_the_boundary_correction():
    changed = 0
    for name in iu.names():
        g = iu.load(name)["gtf"]
        changed += g["0.5,8"] != g["0.5,0"]
    assert changed >= 3


def test_read_cluster_skips_garbage_and_isoform_lines(tmp_path):
    p = tmp_path / "cluster_c_1.tsv"
    p.write_text("#c\t1\t10,20,30\nisoform_0\t1\t11\n5\tr5\tc\t+\t1\t0\tN\t0\t11\t1\t1\n6\tr6\tc\t+\t1\t0\tS\t*\t10\t1\t0\n")
    segments, reads, isos = isoforms.read_cluster(str(p))
    assert segments == {("c", 1): [(10, 20), (20, 30)]} and list(reads) == [5]
    assert isos == {("c", 1, 0, 0): dict(rids={5})}
    p.write_text("#c\t1\t10,20,30\n5\tr5\tc\t+\t1\t0\tN\t0\t1\t1\n")
    with pytest.raises(AssertionError):                    # label string shorter than the segment list (:192-193)
        isoforms.read_cluster(str(p))


def test_isoforms_library_exports_every_declared_symbol():
    text = open(os.path.join(build.INCLUDE, "freddie_isoforms.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(fiso_[a-z_]+)\s*\(", text)))
    assert declared == sorted(isoforms.EXPORTS)
    L = isoforms.load()
    for name in declared:
        assert hasattr(L, name)
    assert L.fiso_abi_version() == 1


def test_isoforms_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(isoforms.IsoformsError, match="no CPU fallback"):
        isoforms.Context(0)
