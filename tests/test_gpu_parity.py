"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

CASES = [
    ("retention", dict(n_reads=200, n_exons=150, rp=0.05), {}),
    ("dense", dict(n_reads=200, n_exons=150, rp=0.0), {}),
    ("long_reads", dict(n_reads=200, n_exons=150, rp=0.05, max_span=0), {}),
    ("r1000", dict(n_reads=1000, n_exons=150, rp=0.05), {}),
    ("ont_sigma3", dict(n_reads=1000, n_exons=150, rp=0.08, jp=0.8, jsd=6.0), dict(sigma=3.0, threshold_rate=0.8)),
    ("refine", dict(n_reads=600, n_exons=60, rp=0.2, max_span=0), dict(min_read_support_outside=1000)),
    ("weights_ends", dict(n_reads=400, n_exons=80, rp=0.1, jp=0.0), dict(ignore_ends=False, max_problem_size=10, variance_factor=1.0)),
    ("tau_one", dict(n_reads=300, n_exons=60, rp=0.1), dict(threshold_rate=1.0)),
    ("mps100", dict(n_reads=300, n_exons=150, rp=0.3, max_span=0), dict(max_problem_size=100)),
    ("mps100_vf9", dict(n_reads=300, n_exons=150, rp=0.3, max_span=0), dict(max_problem_size=100, variance_factor=9.0)),
    ("sigma12", dict(n_reads=400, n_exons=80, rp=0.3, jp=0.5, jsd=4, max_span=0), dict(sigma=12.0, min_read_support_outside=0)),
    # the CLI's parameter bounds (parse_args :104-109)
    ("sigma50", dict(n_reads=300, n_exons=40, rp=0.1), dict(sigma=50.0)),
    ("sigma50_dense", dict(n_reads=200, n_exons=60, rp=0.0), dict(sigma=50.0)),
    ("sigma50_refine", dict(n_reads=600, n_exons=60, rp=0.2, max_span=0), dict(sigma=50.0, min_read_support_outside=1000)),
    ("sigma01", dict(n_reads=300, n_exons=60, rp=0.1, jp=0.5), dict(sigma=0.1)),
    ("sigma01_refine", dict(n_reads=600, n_exons=60, rp=0.2, max_span=0), dict(sigma=0.1, min_read_support_outside=1000)),
    ("tau05", dict(n_reads=300, n_exons=60, rp=0.1), dict(threshold_rate=0.5)),
    ("tau051", dict(n_reads=300, n_exons=60, rp=0.1), dict(threshold_rate=0.51)),
    ("vf001", dict(n_reads=300, n_exons=60, rp=0.1), dict(variance_factor=0.01)),
    ("vf999", dict(n_reads=300, n_exons=60, rp=0.1), dict(variance_factor=9.99)),
    ("mps5", dict(n_reads=300, n_exons=60, rp=0.3, max_span=0), dict(max_problem_size=5)),
]


@pytest.mark.parametrize("name,gen,params", CASES, ids=[c[0] for c in CASES])
def test_single_partition(gpu_ctx, name, gen, params):
    part = util.make_partition(3, **gen)
    o = util.run_oracle(part, params)
    util.run_gpu(gpu_ctx, [part], params)
    rep = util.compare_partitions(gpu_ctx, [part], [o])
    assert rep["y_identical"], "smoothed signal differs from the oracle by %g" % rep["max_y_err"]


def test_batch_of_partitions(gpu_ctx):
    parts = [util.make_partition(i, n_reads=150 + 37 * i, n_exons=40 + 11 * i, rp=0.05 * (i % 3)) for i in range(12)]
    oracles = [util.run_oracle(p) for p in parts]
    util.run_gpu(gpu_ctx, parts)
    util.compare_partitions(gpu_ctx, parts, oracles)


def test_rerun_is_idempotent(gpu_ctx):
    part = util.make_partition(5, n_reads=300, n_exons=100, rp=0.1)
    util.run_gpu(gpu_ctx, [part])
    a = gpu_ctx.download()
    gpu_ctx.run(); gpu_ctx.sync()
    b = gpu_ctx.download()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_config3_slice_against_oracle(gpu_ctx):
    """40 partitions x 1000 reads generated exactly as the config3 bench workload, every tap against the oracle."""
    from freddie_amd import synth
    kw = dict(synth.WORKLOADS["config3"]); kw.pop("n_partitions")
    parts = [util.make_partition(i, **kw) for i in range(40)]
    oracles = [util.run_oracle(p) for p in parts]
    util.run_gpu(gpu_ctx, parts)
    util.compare_partitions(gpu_ctx, parts, oracles)


@pytest.mark.parametrize("by_seen", [False, True], ids=["by-kept-reads", "by-seen-reads"])
def test_wide_problems_solved_whole_with_sixteen_bit_counters(monkeypatch, by_seen):
    """FSEG_FUSE_LANES=1023: batches of 1 000- and 1 600-read partitions (BASELINE config 3 and wider) are solved whole although
    their widest problems see more than 255 reads.  A problem goes to the 16-bit-counter instance of k_solve when it KEEPS more
    than 255 reads (the 1 600-read partitions have such problems); FSEG_WIDE_BY_SEEN=1 sends every problem that SEES more than
    255 there, so that the 16-bit instances are compared with the oracle on many problems, not a few."""
    from freddie_amd import _lib, synth
    monkeypatch.setenv("FSEG_FUSE_LANES", "1023")
    if by_seen:
        monkeypatch.setenv("FSEG_WIDE_BY_SEEN", "1")
    kw = dict(synth.WORKLOADS["config3"]); kw.pop("n_partitions")
    parts = [util.make_partition(i, **kw) for i in range(16)]
    kw["n_reads"] = 1600
    parts += [util.make_partition(100 + i, **kw) for i in range(8)]
    oracles = [util.run_oracle(p) for p in parts]
    ctx = _lib.Context(0)
    try:
        util.run_gpu(ctx, parts)
        assert 255 < ctx.sizes()["max_problem_reads"] <= 1023
        util.compare_partitions(ctx, parts, oracles)
        ctx.run(); ctx.sync()                                # the replay: only the classes that have wide problems launch their instance
        util.compare_partitions(ctx, parts, oracles)
    finally:
        ctx.close()


def test_config5_slice_against_oracle(gpu_ctx):
    """ONT-like error model with sigma=3.0, threshold_rate=0.80 (BASELINE config 5), 12 partitions x 1000 reads."""
    from freddie_amd import synth
    kw = dict(synth.WORKLOADS["config5"]); kw.pop("n_partitions")
    params = dict(sigma=3.0, threshold_rate=0.8)
    parts = [util.make_partition(i, **kw) for i in range(12)]
    oracles = [util.run_oracle(p, params) for p in parts]
    util.run_gpu(gpu_ctx, parts, params)
    util.compare_partitions(gpu_ctx, parts, oracles)


def test_batch_equals_one_by_one(gpu_ctx):
    """Batching is transparent: a partition's result does not depend on what else is in the batch."""
    parts = [util.make_partition(50 + i, n_reads=200 + 50 * i, n_exons=60, rp=0.1) for i in range(8)]
    util.run_gpu(gpu_ctx, parts)
    pfo, fp, lo, lab = gpu_ctx.download()
    for i, p in enumerate(parts):
        util.run_gpu(gpu_ctx, [p])
        a, b, c, d = gpu_ctx.download()
        assert np.array_equal(b, fp[pfo[i]:pfo[i + 1]])
        assert np.array_equal(d, lab[lo[i]:lo[i + 1]])


def test_tiny_problem_path_forced(monkeypatch):
    """k_tiny (one wave solves a problem of <= 8 candidates whole) is normally switched on by the problem count of the
    previous run (> 256); here it is forced for every batch, and the second run of each batch -- the one that uses it --
    is compared with the oracle on every tap: single partitions of every parameter flavour, and batches."""
    from freddie_amd import _lib, synth
    monkeypatch.setenv("FSEG_TINY_FROM", "0")
    ctx = _lib.Context(0)
    try:
        for name, gen, params in CASES:
            part = util.make_partition(3, **gen)
            o = util.run_oracle(part, params)
            util.run_gpu(ctx, [part], params)
            ctx.run(); ctx.sync()
            util.compare_partitions(ctx, [part], [o])
        parts = [util.make_partition(i, n_reads=150 + 37 * i, n_exons=40 + 11 * i, rp=0.05 * (i % 3)) for i in range(12)]
        oracles = [util.run_oracle(p) for p in parts]
        util.run_gpu(ctx, parts)
        ctx.run(); ctx.sync()
        util.compare_partitions(ctx, parts, oracles)
        kw = dict(synth.WORKLOADS["config4"]); kw.pop("n_partitions")
        parts = [util.make_partition(i, **kw) for i in range(30)]
        oracles = [util.run_oracle(p) for p in parts]
        util.run_gpu(ctx, parts)
        ctx.run(); ctx.sync()
        util.compare_partitions(ctx, parts, oracles)
    finally:
        ctx.close()


def test_tiny_path_switches_on_by_itself_and_changes_nothing(monkeypatch):
    """A many-partition batch: the first run sizes the arenas without k_tiny, later runs use it (more than 256 problems);
    a context that never uses it (FSEG_TINY_FROM beyond any problem count) must download the same bytes."""
    from freddie_amd import _lib, synth
    kw = dict(synth.WORKLOADS["config4"]); kw.pop("n_partitions")
    parts = [util.make_partition(100 + i, **kw) for i in range(40)]
    outs = []
    for tiny_from in ("256", "1000000000"):
        monkeypatch.setenv("FSEG_TINY_FROM", tiny_from)
        ctx = _lib.Context(0)
        try:
            util.run_gpu(ctx, parts)
            assert ctx.sizes()["n_problems"] > 256
            for _ in range(2):
                ctx.run(); ctx.sync()
            outs.append(ctx.download())
        finally:
            ctx.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
