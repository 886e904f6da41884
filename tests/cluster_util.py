"""Shared helpers of the clustering pre-ILP tests: fixture loading and random preprocessed tints."""
import gzip
import json
import os
import random

import goldens

CLUSTER_DIR = os.path.join(goldens.GOLDEN_DIR, "cluster")


RANDOM_FIXTURE = "partition_random"


def cluster_names():
    return sorted(f[:-8] for f in os.listdir(CLUSTER_DIR) if f.endswith(".json.gz") and f[:-8] != RANDOM_FIXTURE)


def random_partition_cases():
    """partition_reads() of the reference itself (executed with networkx by tests/golden/make_cluster_golden.py) on seeded
    random_tint() inputs: [dict(seed, n_reps, n_segs, kw, partitions={maximum_ilp_size: [[rids, incomp], ..]})]."""
    return json.loads(gzip.open(os.path.join(CLUSTER_DIR, RANDOM_FIXTURE + ".json.gz")).read().decode())


def canon_partitions(tint):
    """tint['partitions'] in the fixtures' JSON shape."""
    return [[list(rids), [list(pair) for pair in incomp]] for rids, incomp in tint["partitions"]]


def load_cluster(name):
    return json.loads(gzip.open(os.path.join(CLUSTER_DIR, name + ".json.gz")).read().decode())


def segment_tsv_file(name, tmp_path):
    p = os.path.join(str(tmp_path), "segment_%s.tsv" % name)
    open(p, "wb").write(goldens.load(name)["segment_tsv"].tobytes())
    return p


def random_tint(seed, n_reps, n_segs, n_isoforms=6, noise=0.03, tail_p=0.3, empty_p=0.02):
    """A tint as preprocess_ilp() leaves it, without going through files: rows are noisy sub-ranges of a few
    isoform patterns (so the compatibility graph has real structure), some rows all-zero, some with poly tails."""
    rng = random.Random(seed)
    iso = [[1 if rng.random() < 0.6 else 0 for _ in range(n_segs)] for _ in range(n_isoforms)]
    reads, read_reps, I, C, FL = [], [], {}, {}, {}
    for i in range(n_reps):
        row = [0] * n_segs
        if rng.random() >= empty_p:
            pat = iso[rng.randrange(n_isoforms)]
            a = rng.randrange(n_segs); b = rng.randrange(a, n_segs)
            for j in range(a, b + 1):
                row[j] = pat[j] if rng.random() >= noise else 1 - pat[j]
        ones = [j for j, v in enumerate(row) if v == 1]
        lo, hi = (ones[0], ones[-1]) if ones else (-1, n_segs - 1)
        cat = "N"
        r = rng.random()
        if r < tail_p / 2:
            cat, lo = "S", 0
        elif r < tail_p:
            cat, hi = "E", n_segs - 1
        reads.append(dict(id=i, poly_tail_category=cat))
        read_reps.append([i])
        I[i] = row
        C[i] = [1 if (lo <= j <= hi and row[j] == 0) else 0 for j in range(n_segs)]
        FL[i] = (lo, hi)
    return dict(id=seed, chr="c", segs=[(10 * j, 10 * j + 10, 10) for j in range(n_segs)], reads=reads, read_reps=read_reps,
                ilp_data=dict(I=I, C=C, FL=FL, garbage_cost={i: 3 for i in range(n_reps)}))
