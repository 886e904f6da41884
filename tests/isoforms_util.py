"""Helpers of the isoform-consensus tests: fixtures (reference GTFs) and a random stage input."""
import gzip
import json
import os
import random

import goldens

ISO_DIR = os.path.join(goldens.GOLDEN_DIR, "isoforms")


def names():
    return sorted(f[:-8] for f in os.listdir(ISO_DIR) if f.endswith(".json.gz"))


def load(name):
    return json.loads(gzip.open(os.path.join(ISO_DIR, name + ".json.gz")).read().decode())


def settings():
    return [tuple(s) for s in json.load(open(os.path.join(ISO_DIR, "INDEX.json")))["settings"]]


def write_case(name, tmp_path, split_dir_of):
    """cluster dir + split dir of a fixture case; returns (doc, cluster_tsv, split_tsv)."""
    doc = load(name)
    d, contig, tid = split_dir_of(name, tmp_path)
    cdir = os.path.join(str(tmp_path), "cluster_" + name, contig)
    os.makedirs(cdir, exist_ok=True)
    ctsv = os.path.join(cdir, "cluster_%s_%d.tsv" % (contig, tid))
    open(ctsv, "w").write(doc["cluster_tsv"])
    return doc, ctsv, os.path.join(d, contig, "split_%s_%d.tsv" % (contig, tid)), os.path.dirname(cdir), d


def random_job(seed, n_iso, reads_per_iso, n_segs, tail_p=0.3):
    """(isoforms, segments, reads) as read_cluster() + read_split() leave them, without files."""
    rng = random.Random(seed)
    pos = [1000 + 37 * j + rng.randrange(5) for j in range(n_segs + 1)]
    pos = sorted(set(pos))
    while len(pos) < n_segs + 1:
        pos.append(pos[-1] + 30)
    segs = list(zip(pos[:-1], pos[1:]))
    segments = {("c", seed): segs}
    reads, isoforms = {}, {}
    rid = 0
    for iid in range(n_iso):
        pat = [1 if rng.random() < 0.5 else 0 for _ in range(n_segs)]
        key = ("c", seed, 0, iid)
        isoforms[key] = dict(rids=set())
        for _ in range(reads_per_iso if iid else 2):               # the first isoform is too small for any exon (x >= 3)
            a = rng.randrange(n_segs); b = rng.randrange(a, n_segs)
            data = "".join(str((pat[j] if rng.random() > 0.05 else 1 - pat[j]) if a <= j <= b else (2 if rng.random() < 0.02 else 0))
                           for j in range(n_segs))
            r = rng.random()
            tail = "S" if r < tail_p / 2 else ("E" if r < tail_p else "N")
            starts, ends = [], []
            for j in range(n_segs):
                if data[j] == "1" and (j == 0 or data[j - 1] != "1"):
                    starts.append(segs[j][0] + rng.randrange(-12, 13))
                if data[j] == "1" and (j == n_segs - 1 or data[j + 1] != "1"):
                    ends.append(segs[j][1] + rng.randrange(-12, 13))
            if not starts:
                starts, ends = [pos[0]], [pos[0] + 1]
            reads[rid] = dict(rid=rid, rname="r%d" % rid, chrom="c", strand="+", tint=seed, pid=0, tail=tail, iid=iid, data=data,
                              starts=tuple(starts), ends=tuple(ends))
            isoforms[key]["rids"].add(rid)
            rid += 1
    return isoforms, segments, reads
