"""Clustering pre-ILP, CPU side: the host mirror of read_segment() / preprocess_ilp() against fixtures produced by the
reference's own source (tests/golden/make_cluster_golden.py), the oracle's restatement of partition_reads() against
tint['partitions'] written by the reference's own function (same generator) and on hand-made graphs, and the C-ABI
library's symbols.  No GPU."""
import copy
import os
import re

import numpy as np
import pytest

import cluster_util as cu
from freddie_amd import build, cluster_prep
from oracle import cluster_oracle


@pytest.mark.parametrize("name", cu.cluster_names())
def test_read_segment_and_preprocess_match_reference(name, tmp_path):
    want = cu.load_cluster(name)
    tints = cluster_prep.read_segment(cu.segment_tsv_file(name, tmp_path))
    assert len(tints) == 1
    tint = list(tints.values())[0]
    cluster_prep.preprocess_ilp(tint, dict(recycle_model="constant"))
    d = tint["ilp_data"]
    n = len(tint["read_reps"])
    assert (tint["id"], tint["chr"]) == (want["id"], want["chr"])
    assert [list(s) for s in tint["segs"]] == want["segs"]
    assert tint["read_reps"] == want["read_reps"]
    assert ["".join(map(str, d["I"][i])) for i in range(n)] == want["I"]
    assert ["".join(map(str, d["C"][i])) for i in range(n)] == want["C"]
    assert [list(d["FL"][i]) for i in range(n)] == want["FL"]
    assert [d["garbage_cost"][i] for i in range(n)] == want["garbage_cost"]
    for r, w in zip(tint["reads"], want["reads"]):
        assert (r["id"], r["name"], r["chr"], r["strand"], r["tint"]) == (w["id"], w["name"], w["chr"], w["strand"], w["tint"])
        assert "".join(map(str, r["data"])) == w["data"]
        assert sorted([list(k) + [v] for k, v in r["gaps"].items()]) == w["gaps"]
        assert sorted(map(list, r["softclip"].items())) == [list(x) for x in w["softclip"]]
        assert sorted([k, list(v)] for k, v in r["poly_tail"].items()) == w["poly_tail"]
        assert r["poly_tail_category"] == w["poly_tail_category"]


@pytest.mark.parametrize("name", cu.cluster_names())
def test_oracle_partition_reads_matches_reference_on_segment_tsvs(name, tmp_path):
    """The oracle's restatement of partition_reads() against tint['partitions'] as the reference's own function (run with
    networkx at fixture time) left it, for maximum_ilp_size 7 (even split of large components, :259) and 1000."""
    want = cu.load_cluster(name)
    tint = list(cluster_prep.read_segment(cu.segment_tsv_file(name, tmp_path)).values())[0]
    cluster_prep.preprocess_ilp(tint, dict(recycle_model="constant"))
    for size, parts in want["partitions"].items():
        t = copy.deepcopy(tint)
        cluster_oracle.partition_reads(t, int(size))
        assert cu.canon_partitions(t) == parts, "maximum_ilp_size %s" % size


def test_oracle_partition_reads_matches_reference_on_random_tints():
    cases = cu.random_partition_cases()
    assert len(cases) >= 10
    n_multi = 0
    for case in cases:
        tint = cu.random_tint(case["seed"], case["n_reps"], case["n_segs"], **case["kw"])
        for size, parts in case["partitions"].items():
            t = copy.deepcopy(tint)
            cluster_oracle.partition_reads(t, int(size))
            assert cu.canon_partitions(t) == parts, "seed %d, maximum_ilp_size %s" % (case["seed"], size)
            n_multi += len(parts) > 1
    assert n_multi >= 10                       # the fixtures do exercise components, pruning and the even split


def test_read_segment_error_behaviour(tmp_path):
    p = tmp_path / "segment_c_1.tsv"
    p.write_text("#c\t1\t10,20,15\n")                                   # positions not increasing
    with pytest.raises(AssertionError):
        cluster_prep.read_segment(str(p))
    p.write_text("#c\t1\t10,20,30\n0\tr\tc\t+\t1\t1\t\n")             # one label for two segments
    with pytest.raises(AssertionError):
        cluster_prep.read_segment(str(p))
    p.write_text("#c\t1\t10,20,30\n0\tr\tc\t+\t1\t13\t\n")            # label outside [012]: the line does not match
    with pytest.raises(AttributeError):
        cluster_prep.read_segment(str(p))
    p.write_text("#c\t1\t10,20,30\n0\tr\tc\t+\t1\t10\t\n")
    tint = cluster_prep.read_segment(str(p))[1]
    with pytest.raises(AttributeError):                                  # the reference's exons / introns models call
        cluster_prep.preprocess_ilp(tint, dict(recycle_model="exons"))   # .values() on a list (:187-193, :314-316)


def test_rep_key_groups_by_structure(tmp_path):
    p = tmp_path / "segment_c_2.tsv"
    p.write_text("#c\t2\t0,10,20,30,40\n"
                 "0\ta\tc\t+\t2\t1201\t0-3:5,SSC:4,\n"        # key 1001 .0
                 "1\tb\tc\t+\t2\t1001\t0-3:9,ESC:7,\n"        # same key (gap <= 10 -> 0, soft clips are not in the key)
                 "2\tc\tc\t+\t2\t1001\t0-3:11,\n"             # gap > 10: own rep
                 "3\td\tc\t-\t2\t1001\tEA_25:3,ESC:2,\n"      # poly tail, small gap: .E0
                 "4\te\tc\t-\t2\t1001\tET_30:0,\n")           # same key as read 3 (letter = end, not base)
    tint = cluster_prep.read_segment(str(p))[2]
    assert tint["read_reps"] == [[0, 1], [2], [3, 4]]
    cluster_prep.preprocess_ilp(tint, dict(recycle_model="constant"))
    assert tint["ilp_data"]["FL"] == {0: (0, 3), 1: (0, 3), 2: (0, 3)}
    assert [r["poly_tail_category"] for r in tint["reads"]] == ["N", "N", "N", "E", "E"]
    assert tint["reads"][4]["gaps"] == {(3, 4): 3}                       # the rep's dict is shared by its reads (:324)
    assert tint["ilp_data"]["garbage_cost"] == {0: 6, 1: 3, 2: 6}


def test_oracle_compatibility_rules():
    c = cluster_oracle.compatible
    a = (1, 1, 0, 1, 1, 1)
    assert c((a, (0, 5, "N")), (a, (0, 5, "N")))
    assert not c((a, (0, 5, "S")), (a, (0, 5, "E")))                     # tails on different ends
    assert c((a, (0, 5, "S")), (a, (0, 5, "N")))
    assert c((a, (0, 5, "N")), ((1, 0, 1, 1, 1, 1), (0, 5, "N")))        # 2 differences in an overlap of 6
    assert not c((a, (0, 5, "N")), ((1, 0, 1, 0, 1, 1), (0, 5, "N")))    # 3 differences
    assert not c(((1, 1, 0, 0, 0, 0), (0, 1, "N")), ((0, 0, 0, 0, 1, 1), (4, 5, "N")))   # no overlap
    assert c(((1, 1, 1, 0, 0, 0), (0, 2, "N")), ((0, 0, 1, 1, 1, 0), (2, 4, "N")))       # overlap of one, equal
    assert not c(((1, 1, 0, 0, 0, 0), (0, 2, "E")), ((0, 0, 1, 1, 1, 0), (2, 4, "N")))   # overlap of one, differs
    assert not c(((0,) * 6, (-1, 5, "N")), (a, (0, 5, "N")))             # a read that covers nothing


def test_oracle_prune_and_components():
    # path 0-1-2-3 plus triangle 4-5-6 plus isolated 7: every inner path edge lacks a common neighbour and both of its
    # ends have degree 2 -> (1,2) goes in pass 1; then 0-1 and 2-3 are pendant pairs and stay
    nb, passes = cluster_oracle.prune(8, [(0, 1), (1, 2), (2, 3), (4, 5), (5, 6), (4, 6)])
    assert passes == 1
    assert [sorted(s) for s in nb] == [[1], [0], [3], [2], [5, 6], [4, 6], [4, 5], []]
    assert cluster_oracle.connected_components(8, nb) == [[0, 1], [2, 3], [4, 5, 6], [7]]
    # a 4-cycle: no edge has a common neighbour, all degrees 2 -> everything goes at once
    nb, passes = cluster_oracle.prune(4, [(0, 1), (1, 2), (2, 3), (0, 3)])
    assert passes == 1 and all(len(s) == 0 for s in nb)
    assert list(cluster_oracle.split_list_evenly(list(range(7)), 3)) == [[0, 1, 2], [3, 4, 5], [6]]


def test_pack_structures_layout():
    tint = cu.random_tint(3, n_reps=70, n_segs=40)
    uniq = cluster_prep.unique_structures(tint)
    assert uniq == cluster_oracle.unique_data_of(tint)
    pk = cluster_prep.pack_structures([uniq, uniq[:5]])
    n = len(uniq)
    assert pk["row_off"].tolist() == [0, n, n + 5] and pk["n_seg"].tolist() == [40, 40]
    assert pk["bits_off"].tolist() == [0, 2 * n, 2 * n + 10]
    assert pk["adj_off"].tolist() == [0, n * ((n + 63) // 64), n * ((n + 63) // 64) + 5]
    rows = pk["bits"][:2 * n].reshape(n, 2)
    for r, u in enumerate(uniq):
        got = [(int(rows[r, s // 32]) >> (s % 32)) & 1 for s in range(40)]
        assert got == list(u[0][0])
        assert (pk["first"][r], pk["last"][r], "NSE"[pk["tail"][r]]) == u[0][1]


def test_cluster_library_exports_every_declared_symbol():
    text = open(os.path.join(build.INCLUDE, "freddie_cluster.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(fclu_[a-z_]+)\s*\(", text)))
    assert declared == sorted(cluster_prep.EXPORTS)
    L = cluster_prep.load()
    for name in declared:
        assert hasattr(L, name), "libfreddie_cluster.so does not export %s" % name
    assert L.fclu_abi_version() == 1


def test_cluster_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(cluster_prep.ClusterError, match="no CPU fallback"):
        cluster_prep.Context(0)
    with pytest.raises(cluster_prep.ClusterError):
        cluster_prep.partition_reads(cu.random_tint(1, 10, 8), 1000)
