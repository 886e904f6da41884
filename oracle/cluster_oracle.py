"""CPU restatement of the reference's partition_reads() (vpc-ccg/freddie py/freddie_cluster.py:196-274) -- TEST
INFRASTRUCTURE ONLY: imported by tests/ (and tools/cluster_bench.py's cpu_baseline leg), never by the product.

PINNED: tests/golden/cluster/ holds tint['partitions'] as the reference's own partition_reads() and split_list_evenly()
(:112-116) left it -- their source executed unmodified, with the module's own networkx imports (networkx 3.4.2 is in this
image), by tests/golden/make_cluster_golden.py -- for the eleven segment goldens at maximum_ilp_size 7 and 1000 and for
thirteen seeded random tints; tests/test_cluster_host.py checks this file against all of them.  read_segment() and
preprocess_ilp() (:119-172, :277-328) are pinned the same way.

What follows restates the function step by step with a list-of-sets graph in place of networkx.Graph (the reference uses only
add_nodes_from / add_edges_from / edges / neighbors / remove_edges_from / has_edge and connected_components: an undirected
simple graph; components come out in order of their first node in node insertion order).
"""
from math import ceil


def split_list_evenly(l, m):                      # :112-116
    p = ceil(len(l) / m)
    s = ceil(len(l) / p)
    for idx in range(0, p * s, s):
        yield l[idx:idx + s]


def unique_data_of(tint):                         # :203-215
    reads = tint['reads']
    read_reps = tint['read_reps']
    I = tint['ilp_data']['I']
    FL = tint['ilp_data']['FL']
    unique = dict()
    for i in sorted(I.keys()):
        d = (tuple(I[i]), (FL[i][0], FL[i][1], reads[read_reps[i][0]]['poly_tail_category']))
        if d in unique:
            unique[d].append(i)
        else:
            unique[d] = [i]
    return list(unique.items())


def compatible(u1, u2):                           # the body of the pair loop, :219-234
    d1, (f1, l1, t1) = u1
    d2, (f2, l2, t2) = u2
    if t1 != 'N' and t2 != 'N' and t1 != t2:
        return False
    f = max(f1, f2)
    l = min(l1, l2)
    o = l - f + 1
    w = sum(x == y == 1 for x, y in zip(d1[f:l + 1], d2[f:l + 1]))
    if w < 1:
        return False
    d = sum(x != y for x, y in zip(d1[f:l + 1], d2[f:l + 1]))
    return (o > 3 and d < 3) or (1 <= o <= 3 and d == 0)


def compat_edges(unique):                         # :217-234
    N = len(unique)
    return [(i, j) for i in range(N) for j in range(i + 1, N) if compatible(unique[i][0], unique[j][0])]


def prune(N, edges):                              # :236-255; returns (neighbour sets, passes that removed something)
    nb = [set() for _ in range(N)]
    for i, j in edges:
        nb[i].add(j)
        nb[j].add(i)
    passes = 0
    while True:
        remove = []
        for i in range(N):
            for j in nb[i]:
                if i < j:
                    n1, n2 = nb[i], nb[j]
                    if len(n1) == 1 or len(n2) == 1 or len(n1 & n2) > 0:
                        continue
                    remove.append((i, j))
        for i, j in remove:
            nb[i].discard(j)
            nb[j].discard(i)
        if len(remove) == 0:
            break
        passes += 1
    return nb, passes


def connected_components(N, nb):                  # :257 (networkx order: by first node, nodes were added 0..N-1)
    seen = [False] * N
    out = []
    for s in range(N):
        if seen[s]:
            continue
        comp, stack = [], [s]
        seen[s] = True
        while stack:
            v = stack.pop()
            comp.append(v)
            for w in nb[v]:
                if not seen[w]:
                    seen[w] = True
                    stack.append(w)
        out.append(sorted(comp))
    return out


def partition_reads(tint, maximum_ilp_size):      # :196-274 (without the debugging print of :262)
    unique = unique_data_of(tint)
    N = len(unique)
    nb, _ = prune(N, compat_edges(unique))
    tint['partitions'] = list()
    for comp in connected_components(N, nb):
        for c in split_list_evenly(comp, maximum_ilp_size):
            rids = list()
            incomp = list()
            for idx, i in enumerate(c):
                rids.extend(unique[i][1])
                for j in c[idx + 1:]:
                    if j in nb[i]:
                        continue
                    for rid_1 in unique[i][1]:
                        for rid_2 in unique[j][1]:
                            incomp.append((rid_1, rid_2))
            tint['partitions'].append((rids, incomp))
