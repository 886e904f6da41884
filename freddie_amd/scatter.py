"""Static scatter of independent partitions over the GPUs of one node.

Partitions share nothing (reference: one process-pool task per split file,
py/freddie_segment.py:871-876), so there is no collective: every GPU gets a fixed list of
partitions, chosen longest-processing-time-first on an estimated cost, before anything runs.
"""


def lpt_scatter(costs, n_bins):
    """Deterministic LPT: returns n_bins lists of indices into ``costs``."""
    if n_bins <= 0:
        raise ValueError("n_bins must be positive")
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * n_bins
    bins = [[] for _ in range(n_bins)]
    for i in order:
        b = min(range(n_bins), key=lambda j: (loads[j], j))
        bins[b].append(i)
        loads[b] += costs[i]
    for b in bins:
        b.sort()
    return bins


def rank_share(costs, rank, world_size):
    """The partitions of ``rank`` in a world of ``world_size`` GPUs."""
    return lpt_scatter(costs, world_size)[rank]
