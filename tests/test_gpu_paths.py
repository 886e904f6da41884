"""Code paths that ordinary batches do not reach, each forced and checked against the CPU oracle through the C-ABI:
the three-pass scan and the look-back scan's stall fallback, plain launches / no side streams / guessed arenas
(the diagnostic switches every per-kernel profile uses), the 32-bit-count DP (forced, and for real with a problem that
sees more than 65 535 reads), the huge-problem kernels next to k_tiny in one batch, and the full-size per-GPU batches of
the bench workloads (reference: optimize :475-568, get_cumulative_coverage's uint32 :189-192)."""
import numpy as np
import pytest

import util
from freddie_amd import _lib, synth

pytestmark = pytest.mark.gpu


def mixed_batch():
    """A dozen partitions of different shapes: big and tiny DP problems, refinement, long intervals."""
    parts = [util.make_partition(900 + i, n_reads=120 + 90 * i, n_exons=30 + 13 * i, rp=0.04 * (i % 4), max_span=(0 if i % 3 == 0 else 14))
             for i in range(12)]
    return parts, [util.run_oracle(p) for p in parts]


def check_twice(ctx, parts, oracles, params=None):
    """First run of the batch (sized, plain launches unless disabled) and the replayed run must both match."""
    util.run_gpu(ctx, parts, params)
    rep = util.compare_partitions(ctx, parts, oracles)
    assert rep["y_identical"]
    ctx.run(); ctx.sync()
    rep = util.compare_partitions(ctx, parts, oracles)
    assert rep["y_identical"]
    a = ctx.download()
    b = ctx.results()                                       # pinned zero-copy results == copied results
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    c = ctx.results(packed=True)                            # ... and the two-bit form of the labels
    for x, y in zip(a[:3], c[:3]):
        assert np.array_equal(x, y)
    assert np.array_equal(c[3], util.pack_labels(a[3]))
    assert np.array_equal(ctx.results()[3], a[3])           # switching back refetches the byte form


SWITCHES = [
    {"FSEG_SCAN_SINGLE_MAX": "0"},                          # three-pass scan (k_scan1 / k_scan2 + block sums)
    {"FSEG_FORCE_SCAN_STALL": "1"},                         # the look-back scan reports a stall -> rerun with the three-pass scan
    {"FSEG_NO_GRAPH": "1"},
    {"FSEG_NO_FORK": "1"},
    {"FSEG_NO_GRAPH": "1", "FSEG_NO_FORK": "1"},            # how the per-kernel profiles are taken
    {"FSEG_NO_SIZED": "1"},                                 # guessed arenas, overflow -> grow -> re-run (the fallback path)
    {"FSEG_NO_SIZED": "1", "FSEG_FORCE_SCAN_STALL": "1"},
    {"FSEG_TINY_FROM": "0"},                                # k_wave / k_tiny take the small problems of a batch of few
    {"FSEG_NO_FUSE": "1"},                                  # every non-tiny problem through the arena path (tiles, work items, k_score, k_dp*)
    {"FSEG_NO_FUSE": "1", "FSEG_TINY_FROM": "1000000000", "FSEG_NO_SIZED": "1"},
    {"FSEG_TINY_FROM": "1000000000"},                       # ... and the tiny ones through k_solve (no problem count reaches the limit)
    {"FSEG_NO_WAVE": "1"},                                  # k_tiny instead of k_wave<8> (batches with a rep of > 510 exons take this)
    {"FSEG_FUSE_LANES": "255"},                             # batches with a problem that sees more than 255 reads: the arena path
    {"FSEG_FUSE_LANES": "1023"},
    {"FSEG_SCORE_PLAN": "0"},                               # ... all on the main stream (no k_gate)
    {"FSEG_SCORE_PLAN": "BM|gTS"},                          # ... no W: the 16-bit-counter instances behind their classes' 8-bit ones
    {"FSEG_SCORE_PLAN": "B|M|S|g|T"},                       # more segments than streams: not a plan, one stream
    {"FSEG_SCORE_PLAN": "gBMTS"},                           # a gate in front of its own kernel: leaves by its time limit
    {"FSEG_SCORE_PLAN": "B|gM|gS|gT"},                      # ... or each class behind the gate on its own stream
    {"FSEG_FORCE_KEY64": "1"},                              # 64-bit DP keys (batches with a partition of 2^18 reads or more)
    {"FSEG_FORCE_KEY64": "1", "FSEG_NO_WAVE": "1", "FSEG_SCORE_PLAN": "0"},
    {"FSEG_GLOBAL_SORT": "1"},                              # the batch-wide radix sort of the reps instead of the in-LDS sort per partition
    {"FSEG_NO_SDMA_D2H": "1"},                              # results through the runtime's copy instead of the SDMA engine
    {"FSEG_SPLIT_DP": "0"},                                 # the DP as the tail of k_solve's workgroups (one wave: dp_solve_wave), no k_dpw
    {"FSEG_SPLIT_DP": "0", "FSEG_FORCE_KEY64": "1"},        # ... with 64-bit keys: the large class's DP by the whole workgroup (dp_solve_push)
    {"FSEG_SPLIT_DP": "5", "FSEG_SCORE_PLAN": "BM|gTS"},    # small and large class split, the mid class fused; round 3's plan
    {"FSEG_SPLIT_DP": "15"},                                # the large class's k_dpw by one wave a problem (default: eight, dp_solve_push on the handed-over tables)
    {"FSEG_SPLIT_DP": "7", "FSEG_FORCE_KEY64": "1", "FSEG_FUSE_LANES": "1023", "FSEG_WIDE_BY_SEEN": "1"},   # eight-wave k_dpw: 64-bit keys, 16-bit counters
    {"FSEG_NO_FORK": "1", "FSEG_FORCE_KEY64": "1", "FSEG_FUSE_LANES": "1023"},   # k_dpw with 64-bit keys on one stream, the 16-bit instances behind the 8-bit ones
    {"FSEG_FUSE_LANES": "1023", "FSEG_WIDE_BY_SEEN": "1"},  # wide problems (16-bit counters) through the split path, chosen by the reads they see
    {"FSEG_FUSE_LANES": "1023", "FSEG_WIDE_BY_SEEN": "1", "FSEG_SPLIT_DP": "0"},   # ... and fused: the 16-bit instances go over their classes' wide lists either way
    {"FSEG_FUSE_LANES": "1023", "FSEG_WIDE_BY_SEEN": "1", "FSEG_FORCE_KEY64": "1"},
    {"FSEG_FUSE_LANES": "1023", "FSEG_SCORE_PLAN": "gM|B|gTS"},    # no `h` gate, no W
    {"FSEG_THR_PART": "1"},                                 # the variance threshold of a partition by one workgroup (k_thr_part), whatever the batch's size
    {"FSEG_THR_PART": "0"},                                 # ... and never: the batch-wide compaction + a workgroup per 8192-value chunk
    {"FSEG_THR_PART": "1", "FSEG_NO_FORK": "1", "FSEG_NO_GRAPH": "1"},
    {"FSEG_LABEL_BYTES": "1"},                              # the label arena as bytes (the default writes two bits per label unless the threshold table holds a 1.0)
    {"FSEG_DEV_SYNC": "0"},                                 # the scoring stage's side streams forked and joined with events
]


@pytest.mark.parametrize("env", SWITCHES, ids=["+".join("%s=%s" % kv for kv in e.items()) for e in SWITCHES])
def test_diagnostic_switches_keep_parity(env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    parts, oracles = mixed_batch()
    ctx = _lib.Context(0)
    try:
        check_twice(ctx, parts, oracles)
        one = [parts[3]]                                    # a one-partition (small) batch on the same context
        check_twice(ctx, one, [oracles[3]])
    finally:
        ctx.close()


@pytest.mark.parametrize("mode,env", [(1, {}), (2, {}), (3, {}), (1, {"FSEG_NO_FORK": "1"}), (2, {"FSEG_DEV_SYNC": "0"})],
                         ids=["all-stages", "scoring-only", "plain-replays", "one-stream-graphs", "scoring-only-event-joins"])
def test_stage_events_keep_parity(mode, env, monkeypatch):
    """fseg_set_profiling: events around the stages (the scoring stage's begin event doubles as its first fork) change no
    result, and the stage times they give are there on the first run and on replays."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    parts, oracles = mixed_batch()
    ctx = _lib.Context(0)
    try:
        ctx.set_profiling(mode)
        check_twice(ctx, parts, oracles)
        ms = ctx.stage_ms()
        assert ms["interval_scoring"] > 0.0 and all(v >= 0.0 for v in ms.values())
        ctx.run(); ctx.sync()                               # a second replay with the events in place
        assert util.compare_partitions(ctx, parts, oracles)["y_identical"]
        assert ctx.stage_ms()["interval_scoring"] > 0.0
        ctx.set_profiling(0)
        check_twice(ctx, parts, oracles)
    finally:
        ctx.close()


def test_fuse_turning_on_recounts_the_reads_wide_problems_keep(monkeypatch):
    """A batch that is not solved whole (its widest problem sees too many reads) leaves the kept-read counts of k_prob_range
    out; the next batch on the same context, solved whole with 16-bit counters for some problems, must get them (the sized
    run's rescan redoes k_prob_range) -- and the other way round."""
    monkeypatch.setenv("FSEG_FUSE_LANES", "1023")
    deep = [util.make_partition(4242, n_reads=4000, n_exons=14, rp=0.02, max_span=0)]
    deep_oracle = [util.run_oracle(p) for p in deep]
    parts, oracles = mixed_batch()
    ctx = _lib.Context(0)
    try:
        for rnd in range(2):
            util.run_gpu(ctx, deep, None)
            assert ctx.sizes()["max_problem_reads"] > 1023              # the arena path: nothing is solved whole
            assert util.compare_partitions(ctx, deep, deep_oracle)["y_identical"]
            check_twice(ctx, parts, oracles)
            assert 255 < ctx.sizes()["max_problem_reads"] <= 1023      # solved whole, some problems by the 16-bit instances
    finally:
        ctx.close()


def test_wide_count_dp_for_real():
    """One partition whose DP windows see more than 65 535 reads: the 16-bit count tables cannot hold out(i,j,k), the
    library must pick the 32-bit DP by itself."""
    part = util.make_partition(77, n_reads=90000, n_exons=14, rp=0.5, jp=0.8, jsd=6.0, max_span=0)    # 3 problems, one sees 72 896 reads
    o = util.run_oracle(part)
    ctx = _lib.Context(0)
    try:
        util.run_gpu(ctx, [part])
        assert ctx.sizes()["max_problem_reads"] >= 65536, ctx.sizes()
        util.compare_partitions(ctx, [part], [o])
        ctx.run(); ctx.sync()
        util.compare_partitions(ctx, [part], [o])
    finally:
        ctx.close()


def test_huge_problems_and_tiny_problems_in_one_batch(monkeypatch):
    """max_problem_size 100: problems beyond the LDS-resident kernels (k_score_huge / k_dp_huge) together with many
    partitions of tiny problems solved by k_tiny."""
    monkeypatch.setenv("FSEG_TINY_FROM", "0")
    params = dict(max_problem_size=100)
    parts = [util.make_partition(3, n_reads=300, n_exons=150, rp=0.3, max_span=0)]
    parts += [util.make_partition(500 + i, n_reads=200, n_exons=150, rp=0.0) for i in range(6)]      # exon-dense: n <= 4
    oracles = [util.run_oracle(p, params) for p in parts]
    ctx = _lib.Context(0)
    try:
        util.run_gpu(ctx, parts, params)
        sz = ctx.sizes()
        assert sz["max_problem_size"] > 60, sz
        util.compare_partitions(ctx, parts, oracles)
        ctx.run(); ctx.sync()
        util.compare_partitions(ctx, parts, oracles)
    finally:
        ctx.close()


@pytest.mark.parametrize("env", [{}, {"FSEG_NO_SIZED": "1"}], ids=["sized", "guessed-arenas"])
def test_label_arena_follows_the_parameters(env, monkeypatch):
    """threshold_rate < 1 writes the labels at two bits each, threshold_rate = 1 as bytes (a column's default can be '2'): one
    context going back and forth between the two -- first runs and replays, sized and with guessed arenas -- always has the arena
    its run needs."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    parts = [util.make_partition(9100 + i, n_reads=200 + 30 * i, n_exons=50, rp=0.1) for i in range(5)]
    ctx = _lib.Context(0)
    try:
        for tau in (1.0, 0.9, 1.0, 0.8, 0.9, 1.0):
            params = dict(threshold_rate=tau)
            oracles = [util.run_oracle(p, params) for p in parts]
            check_twice(ctx, parts, oracles, params)
    finally:
        ctx.close()


def test_rate_just_below_one_whose_table_ends_at_one():
    """threshold_rate = 0.9999 (parse_args :105 accepts it): smooth_threshold() (:277-286) rounds its entry for segments of 107
    positions to 1.0, so in THAT column every read -- covered or not -- is '2' (:816-828 with h = 1, l = 0) although the rate is
    below 1.  The two-bit label arena assumes no column's default is '2': the library must decide from the table.
    The reference-minted golden e_tau9999_len107 (one segment of exactly 107 positions) inside a batch of ordinary
    partitions, labels as bytes, packed, first run and replay; then the same context back at an ordinary rate."""
    import goldens
    g = goldens.load("e_tau9999_len107")
    params, tabs = goldens.params_of(g), goldens.tables_of(g)
    assert params["threshold_rate"] < 1.0 and tabs["h_table"][107] == 1.0
    gold_part, gold_res = goldens.partition_of(g), goldens.as_oracle_result(g)
    assert (gold_res["labels"][:, 0] == 2).all()                            # the column of length 107: '2' for every rep
    others = [util.make_partition(9300 + i, n_reads=150 + 25 * i, n_exons=40, rp=0.1) for i in range(6)]
    parts = others[:3] + [gold_part] + others[3:]
    oracles = [util.run_oracle(p, params, tabs) for p in others]
    oracles = oracles[:3] + [gold_res] + oracles[3:]
    ctx = _lib.Context(0)
    try:
        check_twice(ctx, parts, oracles, params)
        check_twice(ctx, [gold_part], [gold_res], params)                   # alone (a small batch)
        check_twice(ctx, others, [util.run_oracle(p) for p in others])      # the packed arena again
        check_twice(ctx, parts, oracles, params)
    finally:
        ctx.close()


@pytest.mark.parametrize("mps", [150, 300])
def test_giant_problems_beyond_the_lds_kernels(mps, monkeypatch):
    """max_problem_size 150 / 300 (the CLI, like the reference's parse_args :108, accepts any value > 3; optimize :475-568 has no
    size limit): DP problems of 130 .. 300 candidates go through k_score_giant / k_dp_giant (every per-pair table in global
    scratch), next to ordinary partitions in the same batch, sized first run and replay, and with guessed arenas."""
    params = dict(max_problem_size=mps, variance_factor=9.99)
    parts = [util.make_partition(77, n_reads=300, n_exons=400, rp=0.3, max_span=0), util.make_partition(78, n_reads=200, n_exons=600, rp=0.5, max_span=0)]
    parts += [util.make_partition(600 + i, n_reads=150, n_exons=60, rp=0.05) for i in range(4)]
    oracles = [util.run_oracle(p, params) for p in parts]
    sizes = np.concatenate([o["prob_end"] - o["prob_start"] + 1 for o in oracles])
    assert sizes.max() > 128 and (mps == 150 or sizes.max() > 256), sizes.max()
    for env in ({}, {"FSEG_NO_SIZED": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = _lib.Context(0)
        try:
            check_twice(ctx, parts, oracles, params)
            assert ctx.sizes()["max_problem_size"] == sizes.max()
            small = parts[2:]                                     # the next batch has no such problem
            check_twice(ctx, small, oracles[2:], params)
        finally:
            ctx.close()


def test_problems_beyond_the_largest_kernel_are_refused(gpu_ctx):
    """More than 1 024 candidates in one problem (max_problem_size beyond what the CLI lets through): FSEG_ERR_UNSUPPORTED, no labels."""
    params = dict(max_problem_size=5000, variance_factor=9.99)
    part = util.make_partition(79, n_reads=100, n_exons=1500, rp=0.6, max_span=0)
    with pytest.raises(_lib.SegError, match="more than 1024 candidates"):
        util.run_gpu(gpu_ctx, [part], params)
        gpu_ctx.download()
    good = util.make_partition(5, n_reads=100, n_exons=20)
    util.run_gpu(gpu_ctx, [good])                                 # the context is still usable
    util.compare_partitions(gpu_ctx, [good], [util.run_oracle(good)])


@pytest.mark.parametrize("workload,n_part", [("config4", 500), ("config3", 250), ("config5", 250)])
def test_full_size_bench_batches_against_oracle(workload, n_part, gpu_ctx):
    """The exact batches bench.py times (config4: 500 x 500 reads; config3: 250 x 1000 reads; config5: 250 x 1000 reads of the
    ONT-like error model under sigma = 3, threshold rate 0.8), every tap of every partition against the oracle."""
    kw = dict(synth.WORKLOADS[workload]); kw.pop("n_partitions")
    params = dict(sigma=3.0, threshold_rate=0.8) if workload == "config5" else None
    parts = [util.make_partition(i, **kw) for i in range(n_part)]
    oracles = [util.run_oracle(p, params) for p in parts]
    util.run_gpu(gpu_ctx, parts, params)
    rep = util.compare_partitions(gpu_ctx, parts, oracles)
    assert rep["y_identical"]
    gpu_ctx.run(); gpu_ctx.sync()
    util.compare_partitions(gpu_ctx, parts, oracles)


@pytest.mark.parametrize("env,waiters,still_on", [({}, True, True), ({"FSEG_DEV_SYNC": "0"}, False, False), ({"FSEG_SYNC_TICKS": "1"}, True, False),
                                                  ({"FSEG_SCORE_PLAN": "gM|hB|gS|T"}, True, True), ({"FSEG_NO_SIZED": "1"}, True, True), ({"FSEG_NO_SIZED": "1", "FSEG_DEV_SYNC": "0"}, False, False)],
                         ids=["device-side", "events", "waiter-times-out", "three-side-streams", "unsized-first-run", "unsized-first-run-events"])
def test_scoring_stage_fork_and_join(env, waiters, still_on, monkeypatch):
    """The scoring stage's side streams (FSEG_SCORE_PLAN) are forked and joined on the device (k_wait_word / k_signal behind
    k_prob_emit's last workgroup) or with events (FSEG_DEV_SYNC=0); a waiter that gives up (forced: one tick) must end the
    stage unharmed -- the kernels behind it do nothing -- and the batch is rerun with events.  Either way: the oracle's results,
    first run and replays.  (A batch large enough for a plan: many problems in every size class, one context alone.)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    kw = dict(synth.WORKLOADS["config4"]); kw.pop("n_partitions")
    parts = [util.make_partition(7000 + i, **kw) for i in range(48)]
    oracles = [util.run_oracle(p) for p in parts]
    ctx = _lib.Context(0)
    try:
        check_twice(ctx, parts, oracles)
        for _ in range(3):
            ctx.run(); ctx.sync()
        util.compare_partitions(ctx, parts, oracles)
        sy = ctx.tap("sync")
        assert (sy[0] > 0) == waiters, sy                      # stages enqueued with device-side waiters
        assert bool(sy[1]) == still_on, sy
        if waiters and still_on:
            assert sy[2] == sy[0] and sy[3] == sy[0] and sy[5] == 0, sy     # the words carry the last generation; the counter is back at 0
            assert sy[6] == 0, sy                                          # no waiter reached its limit (10 ms per 2^18 reads)
        if waiters and not still_on:
            assert sy[6] >= 3, sy                                          # every forked run lost its waiter, was redone with events; after three: events for good
        assert sy[7] >= 5, sy                                              # every run of a context alone owns the device
        more = [util.make_partition(7100 + i, **kw) for i in range(40)]    # another batch on the same context
        check_twice(ctx, more, [util.run_oracle(p) for p in more])
    finally:
        ctx.close()


@pytest.mark.parametrize("mode", ["0", "1"], ids=["chunk-kernels", "workgroup-per-partition"])
def test_threshold_paths_on_the_goldens_and_big_partitions(mode, monkeypatch):
    """The variance threshold (:757-759, numpy's summation order) both ways -- a workgroup per partition (k_thr_part) and the
    batch-wide compaction with a workgroup per chunk -- on every golden of the reference (NaN threshold, sigma 50, ...), on a
    partition with dozens of 8192-value chunks, and on a many-partition batch; bit-identical thresholds either way."""
    import goldens
    monkeypatch.setenv("FSEG_THR_PART", mode)
    ctx = _lib.Context(0)
    try:
        for name in goldens.names():
            if name == "g4_config2":
                continue
            g = goldens.load(name)
            part = goldens.partition_of(g)
            util.run_gpu(ctx, [part], goldens.params_of(g), goldens.tables_of(g))
            util.compare_partitions(ctx, [part], [goldens.as_oracle_result(g)])
        big = [util.make_partition(8100, n_reads=20000, n_exons=400, rp=0.05), util.make_partition(8101, n_reads=300, n_exons=30)]
        oracles = [util.run_oracle(p) for p in big]
        assert int((oracles[0]["Y"] > 0).sum()) > 4 * 8192                       # several chunks, the last one partial
        check_twice(ctx, big, oracles)
        kw = dict(synth.WORKLOADS["config4"]); kw.pop("n_partitions")
        parts = [util.make_partition(8200 + i, **kw) for i in range(70)]          # (64 partitions and more: the default takes k_thr_part)
        check_twice(ctx, parts, [util.run_oracle(p) for p in parts])
    finally:
        ctx.close()


@pytest.mark.parametrize("big", [False, True], ids=["sorted-in-LDS", "batch-wide-sort"])
def test_lane_preparation_on_the_device(big, gpu_ctx):
    """fseg_upload's device-side ordering of the reads: every rep repeated rep_weight times, sorted inside its partition by
    first position (ties in rep order), with the running maximum of the last positions.  With a partition of more than 2 048
    reps the batch goes through the radix sort and the block-parallel lane kernels (k_lane_blocks / _block_scan / _emit)."""
    parts = [util.make_partition(40 + i, n_reads=300 + 50 * i, n_exons=40, rp=0.1, dedupe=True) for i in range(5)]
    if big:
        parts.insert(2, util.make_partition(46, n_reads=9000, n_exons=60, rp=0.2, dedupe=True))
        assert parts[2].n_reps > 2048
    util.run_gpu(gpu_ctx, parts)
    start, pmax, exons = gpu_ctx.tap("lane_start"), gpu_ctx.tap("lane_pmax"), gpu_ctx.tap("lane_exons")
    lstream, estream = gpu_ctx.tap("lane_stream"), gpu_ctx.tap("exon_stream")
    l0 = e0 = 0
    for p in parts:
        first = p.ex_ts[p.rep_exon_off[:-1]]
        last = p.ex_te[p.rep_exon_off[1:] - 1]
        order = np.lexsort((np.arange(p.n_reps), first))
        lanes = np.repeat(order, p.rep_weight[order])
        n = len(lanes)
        assert n == p.rep_weight.sum()
        assert np.array_equal(start[l0:l0 + n], first[lanes])
        assert np.array_equal(pmax[l0:l0 + n], np.maximum.accumulate(last[lanes]))
        assert np.array_equal(exons[l0:l0 + n, 0], p.rep_exon_off[lanes] + e0)
        assert np.array_equal(exons[l0:l0 + n, 1], p.rep_exon_off[lanes + 1] + e0)
        # the exon stream: the partition's exons once more as (ts, te) pairs, rep after rep in lane order; a lane's range
        # in it holds exactly its rep's exons (the copies of a weighted rep share one range)
        n_ex = np.diff(p.rep_exon_off)[order]
        off = np.concatenate([[0], np.cumsum(n_ex)]) + e0
        rep_pos = np.repeat(np.arange(p.n_reps), p.rep_weight[order])
        assert np.array_equal(lstream[l0:l0 + n, 0], off[rep_pos])
        assert np.array_equal(lstream[l0:l0 + n, 1], off[rep_pos + 1])
        idx = np.concatenate([np.arange(p.rep_exon_off[r], p.rep_exon_off[r + 1]) for r in order])
        assert np.array_equal(estream[e0:e0 + len(idx), 0], p.ex_ts[idx])
        assert np.array_equal(estream[e0:e0 + len(idx), 1], p.ex_te[idx])
        l0 += n; e0 += len(p.ex_ts)
    assert l0 == len(start)


def test_upload_rejects_what_read_split_asserts(gpu_ctx):
    """The per-read assertions of read_split() / process_splicing_data (:158-161, :666-668) are checked on the device
    during the upload; the batch must be refused by the first call that waits."""
    good = util.make_partition(5, n_reads=100, n_exons=20)
    from freddie_amd import pack
    cases = []
    p = pack.PackedPartition(good.iv_start.copy(), good.iv_end.copy(), good.rep_weight.copy(), good.rep_exon_off.copy(),
                             good.ex_ts.copy(), good.ex_te.copy(), good.read_rep)
    p.ex_te[3] = p.ex_ts[3]                                       # exon with start >= end
    cases.append((p, "start >= end"))
    p = pack.PackedPartition(good.iv_start.copy(), good.iv_end.copy(), good.rep_weight.copy(), good.rep_exon_off.copy(),
                             good.ex_ts.copy(), good.ex_te.copy(), good.read_rep)
    e = int(p.rep_exon_off[1])                                    # second exon of rep 0 moved before the first
    if e >= 2:
        p.ex_ts[1], p.ex_te[1] = p.ex_ts[0] - 50, p.ex_ts[0] - 10
        cases.append((p, "out of order|inside one tint interval"))
    p = pack.PackedPartition(good.iv_start.copy(), good.iv_end.copy(), good.rep_weight.copy(), good.rep_exon_off.copy(),
                             good.ex_ts.copy(), good.ex_te.copy(), good.read_rep)
    p.ex_te[int(p.rep_exon_off[-1]) - 1] = int(p.iv_end[-1]) + 5  # runs past the last tint interval
    cases.append((p, "inside one tint interval"))
    for bad, msg in cases:
        with pytest.raises(_lib.SegError, match=msg):
            util.run_gpu(gpu_ctx, [good, bad])
            gpu_ctx.download()
    util.run_gpu(gpu_ctx, [good])                                 # the context is still usable
    util.compare_partitions(gpu_ctx, [good], [util.run_oracle(good)])


@pytest.mark.parametrize("env", [{}, {"FSEG_TINY_FROM": "1000000000"}, {"FSEG_TINY_FROM": "0"}], ids=["default", "no-k_tiny", "k_tiny"])
def test_windows_wider_than_sixteen_bits(env, monkeypatch):
    """Problems whose window spans more than 65 535 positions (a long unspliced interval): coverage and thresholds beyond
    16 bits, in every size class, alone and inside a larger batch.  (16-bit coverage rows in the fused solver were tried for
    their smaller LDS footprint and lost 5 %; this is the case they would have needed a second path for.)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    wide = [util.wide_window_partition(s, per_cluster=pc) for s, pc in ((0, 4), (1, 8), (2, 16), (3, 28))]
    oracles = [util.run_oracle(p) for p in wide]
    spans = []
    for o in oracles:
        c = o["cands"]
        spans.append(max((int(c[e] - c[s]), int(e - s + 1)) for s, e in zip(o["prob_start"], o["prob_end"])))
    assert all(sp[0] >= 65536 for sp in spans), spans
    assert max(sp[1] for sp in spans) > 32 and min(sp[1] for sp in spans) <= 8, spans       # every size class has a wide problem
    ctx = _lib.Context(0)
    try:
        check_twice(ctx, [wide[1]], [oracles[1]])                       # a small batch: one launch for every class
        parts, more = mixed_batch()
        check_twice(ctx, parts + wide, more + oracles)                  # the per-class launches
    finally:
        ctx.close()


def test_contexts_taking_turns_on_one_device():
    """What the CLI and the benchmark do: several contexts of one device, one host thread each, batches in flight at the same
    time (a context then keeps to its one stream; the later contexts have no side streams at all; the label matrix leaves
    through the SDMA engine from several threads).  Every batch must come back as if it had run alone."""
    import threading
    n_ctx, rounds = 4, 3
    batches = []
    for b in range(n_ctx * rounds):
        # (the middle round's batches are large enough -- more than 1 MB of packed labels -- to leave through the SDMA engine)
        parts = [util.make_partition(3000 + 17 * b + i, n_reads=150 + 40 * ((b + i) % 5), n_exons=40 + 10 * (i % 4), rp=0.05 * (i % 3))
                 for i in range(6)] if b // n_ctx != 1 else \
                [util.make_partition(3400 + 31 * b + i, n_reads=500, n_exons=150, rp=0.05) for i in range(36)]
        batches.append((parts, [util.run_oracle(p) for p in parts]))
    ctxs = [_lib.Context(0) for _ in range(n_ctx)]
    errors = []

    def worker(k):
        try:
            for r in range(rounds):
                parts, oracles = batches[r * n_ctx + k]
                util.run_gpu(ctxs[k], parts)
                rep = util.compare_partitions(ctxs[k], parts, oracles)
                assert rep["y_identical"]
                packed = ctxs[k].results(packed=True)
                assert np.array_equal(packed[3], util.pack_labels(ctxs[k].download()[3]))
        except BaseException as exc:                              # noqa: BLE001
            errors.append(exc)

    try:
        th = [threading.Thread(target=worker, args=(k,)) for k in range(n_ctx)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errors:
            raise errors[0]
    finally:
        for c in ctxs:
            c.close()


def test_eight_contexts_replaying_resident_batches():
    """The configuration bench.py's `value` runs (the pool map :871-876 as contexts of one device): eight contexts, each with its own
    RESIDENT batch of config4-shaped partitions (64 and more each: k_thr_part and the split DP are on), uploaded and run once, then
    replayed twenty times by eight host threads at once -- everything recomputed per pass, results left in HBM.  Afterwards every tap
    of every context against the oracle.  Only the context that claimed the device may fork (side streams, device-side waiters);
    no waiter may reach its time limit (round 5: four 20 ms time-outs in one such run, two contexts' waiters in front of each other)."""
    import threading
    n_ctx, passes = 8, 20
    kw = dict(synth.WORKLOADS["config4"]); kw.pop("n_partitions")
    batches = []
    for k in range(n_ctx):
        parts = [util.make_partition(61000 + 100 * k + i, **kw) for i in range(64 + 2 * k)]
        batches.append((parts, [util.run_oracle(p) for p in parts]))
    ctxs = [_lib.Context(0) for _ in range(n_ctx)]
    errors = []
    start = threading.Barrier(n_ctx)

    def worker(k):
        try:
            util.run_gpu(ctxs[k], batches[k][0])              # upload + first (sized) run
            start.wait()
            for _ in range(passes):
                ctxs[k].run(); ctxs[k].sync()
        except BaseException as exc:                          # noqa: BLE001
            errors.append(exc)
            start.abort()

    try:
        th = [threading.Thread(target=worker, args=(k,)) for k in range(n_ctx)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errors:
            raise errors[0]
        forked = 0
        for k in range(n_ctx):
            rep = util.compare_partitions(ctxs[k], *batches[k])
            assert rep["y_identical"]
            packed = ctxs[k].results(packed=True)
            assert np.array_equal(packed[3], util.pack_labels(ctxs[k].download()[3]))
            sy = ctxs[k].tap("sync")
            assert sy[6] == 0, "context %d: %d runs lost a waiter to its time limit (%r)" % (k, sy[6], sy)
            forked += int(sy[7])
        assert forked <= n_ctx * (passes + 1)
        # ... and once more one after the other, each context alone on the device (it owns it: forked run, device-side waiters)
        for k in range(n_ctx):
            before = ctxs[k].tap("sync")
            ctxs[k].run(); ctxs[k].sync()
            after = ctxs[k].tap("sync")
            assert after[7] == before[7] + 1 and after[6] == 0, (before, after)
            assert util.compare_partitions(ctxs[k], *batches[k])["y_identical"]
    finally:
        for c in ctxs:
            c.close()


def test_results_outlive_the_next_upload_and_run(gpu_ctx):
    """include/freddie_seg.h: the zero-copy result views stay valid until the next fseg_results*() call on the context --
    fseg_upload + fseg_run of ANOTHER batch leave them alone (the CLI's pipeline relies on it: its writer thread reads
    batch i's views while batch i + 1 is uploaded and run on the same context, freddie_amd/segment.py)."""
    a = [util.make_partition(5100 + i, n_reads=400, n_exons=60, rp=0.05) for i in range(8)]
    b = [util.make_partition(5200 + i, n_reads=900, n_exons=120, rp=0.1) for i in range(12)]     # larger: every slab is regrown
    util.run_gpu(gpu_ctx, a)
    views = gpu_ctx.results(packed=True)
    kept = [np.array(v, copy=True) for v in views]
    want_labels = util.pack_labels(gpu_ctx.download()[3])
    assert np.array_equal(kept[3], want_labels)
    util.run_gpu(gpu_ctx, b)                                      # upload + run + sync of another batch; no result call
    for v, k in zip(views, kept):
        assert np.array_equal(v, k)                               # the retained views still hold batch a's results
    oracles = [util.run_oracle(p) for p in b]
    util.compare_partitions(gpu_ctx, b, oracles)                  # (fseg_download copies into caller memory: views untouched)
    for v, k in zip(views, kept):
        assert np.array_equal(v, k)
    fresh = gpu_ctx.results(packed=True)                          # ... and this call is what replaces them
    assert np.array_equal(fresh[3], util.pack_labels(gpu_ctx.download()[3]))
