"""Clustering pre-ILP on the GPU (row N3): the compatibility graph, its pruning and partition_reads() through the
C-ABI (include/freddie_cluster.h) against the CPU oracle's restatement of py/freddie_cluster.py:196-274 and against
tint['partitions'] as the reference's own partition_reads() wrote it (fixtures), on the reference's own segment TSVs and
on seeded random tints.  Bit-exact: the graph is integer work."""
import copy

import numpy as np
import pytest

import cluster_util as cu
from freddie_amd import cluster_prep
from oracle import cluster_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = cluster_prep.Context(0)
    yield c
    c.close()


def oracle_matrix(n, edges_or_nb):
    A = np.zeros((n, n), bool)
    if edges_or_nb and isinstance(edges_or_nb[0], set):
        for i, s in enumerate(edges_or_nb):
            for j in s:
                A[i, j] = True
    else:
        for i, j in edges_or_nb:
            A[i, j] = A[j, i] = True
    return A


def check_graphs(ctx, tints):
    uniq = [cluster_prep.unique_structures(t) for t in tints]
    packed = cluster_prep.pack_structures(uniq)
    raw, _ = ctx.compat_graph(packed, prune=False)
    pruned, rounds = ctx.compat_graph(packed, prune=True)
    for t, u in enumerate(uniq):
        n = len(u)
        edges = cluster_oracle.compat_edges(u)
        got_raw = cluster_prep.adjacency_matrix(raw, packed, t)
        assert np.array_equal(got_raw, oracle_matrix(n, edges)), "tint %d: compatibility graph differs" % t
        nb, passes = cluster_oracle.prune(n, edges)
        got = cluster_prep.adjacency_matrix(pruned, packed, t)
        assert np.array_equal(got, got.T) and not got.diagonal().any()
        assert np.array_equal(got, oracle_matrix(n, nb) if n else got), "tint %d: pruned graph differs" % t
        assert rounds[t] == passes
    return uniq


@pytest.mark.parametrize("name", cu.cluster_names())
def test_partition_reads_on_reference_segment_tsvs(ctx, name, tmp_path):
    tint = list(cluster_prep.read_segment(cu.segment_tsv_file(name, tmp_path)).values())[0]
    cluster_prep.preprocess_ilp(tint, dict(recycle_model="constant"))
    check_graphs(ctx, [tint])
    want = copy.deepcopy(tint)
    cluster_oracle.partition_reads(want, 1000)
    ref = cu.load_cluster(name)["partitions"]                     # what the reference's own partition_reads() produced
    big = copy.deepcopy(tint)
    cluster_prep.partition_reads(big, 1000, ctx=ctx, verbose=False)
    assert big["partitions"] == want["partitions"]
    assert cu.canon_partitions(big) == ref["1000"]
    small = copy.deepcopy(tint); want_small = copy.deepcopy(want)
    cluster_oracle.partition_reads(want_small, 7)                 # forces the even split of large components (:259)
    cluster_prep.partition_reads(small, 7, ctx=ctx, verbose=False)
    assert small["partitions"] == want_small["partitions"]
    assert cu.canon_partitions(small) == ref["7"]


def test_partition_reads_on_random_tints_against_the_reference(ctx):
    """The GPU path against tint['partitions'] of the reference's own function (networkx) on seeded random tints."""
    for case in cu.random_partition_cases():
        tint = cu.random_tint(case["seed"], case["n_reps"], case["n_segs"], **case["kw"])
        for size, parts in case["partitions"].items():
            t = copy.deepcopy(tint)
            cluster_prep.partition_reads(t, int(size), ctx=ctx, verbose=False)
            assert cu.canon_partitions(t) == parts, "seed %d, maximum_ilp_size %s" % (case["seed"], size)


@pytest.mark.parametrize("rank_switch,prune_lds", [("1", "1"), ("0", "1"), ("1", "0"), ("1", "0,edges=0"), ("1", "words=300")],
                         ids=["rank-tables", "masked-sums", "per-pass-edge-walk", "per-pass-or-of-rows", "both-prunings-in-one-batch"])
def test_random_tints_batched(ctx, rank_switch, prune_lds, monkeypatch):
    """Shapes that cross every tile edge: N around 64 / 128, M around 32 / 64, one-row and empty-ish tints, in one batch.
    Both forms of k_compat (FCLU_RANK=0: a range mask per word; default: rank tables), whose tiles lie on and above the
    diagonal only -- the graphs are compared whole, both triangles.  Both prunings (:240-255): a small tint whole in one workgroup's LDS
    (k_prune_lds, the default), the per-pass kernels (FCLU_PRUNE_LDS=0), and a batch whose tints are divided between them."""
    monkeypatch.setenv("FCLU_RANK", rank_switch)
    if prune_lds.startswith("words="):
        monkeypatch.setenv("FCLU_PRUNE_LDS_WORDS", prune_lds[6:])
    else:
        monkeypatch.setenv("FCLU_PRUNE_LDS", prune_lds[0])
        if prune_lds.endswith("edges=0"):
            monkeypatch.setenv("FCLU_PRUNE_EDGES", "0")             # the per-pass kernels' older form: OR of the neighbours' rows
    shapes = [(1, 5), (2, 1), (63, 31), (64, 32), (65, 33), (130, 64), (200, 65), (257, 100), (40, 300)]
    tints = [cu.random_tint(100 + k, n, m) for k, (n, m) in enumerate(shapes)]
    tints.append(cu.random_tint(200, 150, 20, n_isoforms=2, noise=0.0, tail_p=0.0))      # dense graph: heavy pruning input
    tints.append(cu.random_tint(201, 120, 24, n_isoforms=12, noise=0.1, tail_p=0.6))     # sparse graph, many tails
    check_graphs(ctx, tints)
    want = copy.deepcopy(tints)
    for t in want:
        cluster_oracle.partition_reads(t, 50)
    cluster_prep.partition_reads_batch(tints, 50, ctx, verbose=False)
    for a, b in zip(tints, want):
        assert a["partitions"] == b["partitions"]


def test_partitions_are_a_cover_and_incompatible_pairs_are_non_edges(ctx):
    """Size-independent properties on a tint the Python oracle would take minutes for (N = 1500 unique reads)."""
    tint = cu.random_tint(7, 1500, 90, n_isoforms=10)
    uniq = cluster_prep.unique_structures(tint)
    packed = cluster_prep.pack_structures([uniq])
    adj, _ = ctx.compat_graph(packed, prune=True)
    A = cluster_prep.adjacency_matrix(adj, packed, 0)
    assert np.array_equal(A, A.T) and not A.diagonal().any()
    # a fixed point of the pruning rule: every surviving edge has a pendant end or a common neighbour
    deg = A.sum(1)
    common = (A.astype(np.int32) @ A.astype(np.int32)) > 0
    assert not (A & ~((deg[:, None] == 1) | (deg[None, :] == 1) | common)).any()
    # spot-check the raw graph on random pairs against the oracle's rule
    raw, _ = ctx.compat_graph(packed, prune=False)
    R = cluster_prep.adjacency_matrix(raw, packed, 0)
    rng = np.random.default_rng(1)
    for i, j in rng.integers(0, len(uniq), (3000, 2)):
        if i != j:
            assert R[i, j] == cluster_oracle.compatible(uniq[i][0], uniq[j][0])
    assert not (A & ~R).any()                                     # pruning only removes
    cluster_prep.partition_reads(tint, 400, ctx=ctx, verbose=False)
    rids = sorted(r for part, _ in tint["partitions"] for r in part)
    assert rids == sorted(tint["ilp_data"]["I"].keys())          # every rep in exactly one partition
    assert all(len(part) <= 400 for part, _ in tint["partitions"])
    rep_row = {rid: k for k, (_, members) in enumerate(uniq) for rid in members}
    for part, incomp in tint["partitions"]:
        s = set(part)
        for a, b in incomp[:2000]:
            assert a in s and b in s and not A[rep_row[a], rep_row[b]]


def test_rows_longer_than_the_rank_tables_hold(ctx):
    """A tint of 7 000 segments (219 words a row: beyond kRankWords = 207) takes the masked form whatever the switch says; a batch
    that mixes it with short rows does so whole."""
    tints = [cu.random_tint(300, 90, 7000, n_isoforms=5), cu.random_tint(301, 70, 40)]
    check_graphs(ctx, tints)


def test_rejects_bad_shapes(ctx):
    uniq = cluster_prep.unique_structures(cu.random_tint(5, 10, 12))
    packed = cluster_prep.pack_structures([uniq])
    bad = dict(packed); bad["last"] = packed["last"].copy(); bad["last"][0] = 12          # beyond the last segment
    with pytest.raises(cluster_prep.ClusterError, match="out of range"):
        ctx.compat_graph(bad)
    bad = dict(packed); bad["first"] = packed["first"].copy()
    k = int(np.argmax(packed["last"] > packed["first"]))
    bad["first"][k] += 1                                                                  # the read's first covered segment now lies in front of `first`
    with pytest.raises(cluster_prep.ClusterError, match="outside"):
        ctx.compat_graph(bad)
    bad = dict(packed); bad["adj_off"] = packed["adj_off"].copy(); bad["adj_off"][1] += 1
    with pytest.raises(cluster_prep.ClusterError, match="adj_off"):
        ctx.compat_graph(bad)
