"""CPU restatement of the per-read loops of the reference's isoform-consensus stage (vpc-ccg/freddie
py/freddie_isoforms.py: isoforms_cons :203-250, correct_boundaries :122-140) -- TEST INFRASTRUCTURE ONLY, imported by
tests/ and tools/isoforms_bench.py's cpu_baseline leg, never by the product.

Pinned: py/freddie_isoforms.py needs only the standard library, so it was imported in the build container and run on
synthetic cluster files; tests/golden/isoforms/ holds its GTF output (tests/golden/make_isoforms_golden.py), and
tests/test_isoforms_host.py checks that these functions reproduce it byte for byte."""
from itertools import groupby


def isoforms_cons(isoforms, segments, reads):
    for isoform_key, isoform in isoforms.items():
        chrom, tint, _, _ = isoform_key
        segs = segments[(chrom, tint)]
        M = len(segs)
        cons = [0] * M
        cov = [0] * M
        tails = {'N': 0, 'S': 0, 'E': 0}
        for rid in isoform['rids']:
            read = reads[rid]
            assert len(read['data']) == M
            if '1' not in read['data']:
                continue                                               # :215-216
            if read['tail'] == 'S':                                    # :217-224 (both tests are on 'S' in the reference)
                first, last = 0, M - 1
            else:
                first = read['data'].index('1')
                last = M - 1 - read['data'][::-1].index('1')
            assert 0 <= first <= last < M
            for j in range(first, last + 1):
                cons[j] += read['data'][j] == '1'
                cov[j] += 1
            tails[read['tail']] += 1
        flags = [x / c > 0.5 if x >= 3 else False for x, c in zip(cons, cov)]   # :233
        if True not in flags:
            continue
        isoform['strand'] = '-' if tails['S'] > tails['E'] else '+'   # :236-239
        starts, ends = [], []
        for d, group in groupby(enumerate(flags), lambda x: x[1]):     # :242-248
            if d != True:
                continue
            group = list(group)
            starts.append(segs[group[0][0]][0])
            ends.append(segs[group[-1][0]][1])
        isoform['starts'], isoform['ends'] = starts, ends
        for s, e in zip(starts, ends):
            assert s < e


def correct_boundaries(side, isoforms, reads, majority_threshold, correction_window):
    if correction_window == 0:
        return
    assert side in ['starts', 'ends']
    for isoform in isoforms.values():
        if side not in isoform:
            continue
        for idx, iso_s in enumerate(isoform[side]):
            cur = {x: 0 for x in range(-correction_window, correction_window + 1)}
            for rid in isoform['rids']:
                for read_s in reads[rid][side]:
                    x = read_s - iso_s
                    if x in cur:
                        cur[x] += 1
            for x, v in cur.items():
                if v / len(isoform['rids']) >= majority_threshold:
                    isoform[side][idx] = x + iso_s
