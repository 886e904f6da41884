"""Maximum sizes (``-m gpu``): the 2 M-read job as ONE batch -- 4 000 partitions, 2.3e8 positions (a tenth of the 2^31 an upload takes),
1.2 M candidates, 100 k problems, 6e8 label bytes: every 64-bit offset, every grid cap and grid-stride loop the 250 k-read batches of
the bench never reach.  Every tap of every partition against the oracle (run partition by partition while the comparison goes on: the
oracle's results of the whole job would be 4 GB), first run and the replay with known sizes.  Reference: the reference takes partitions
one process each (py/freddie_segment.py:871-876); batching is this library's, so its limits are this library's to test."""
import pytest

import util
from freddie_amd import _lib, synth

pytestmark = pytest.mark.gpu


def test_the_whole_job_as_one_batch():
    kw = dict(synth.WORKLOADS["config4"]); n_part = kw.pop("n_partitions")
    parts = [util.make_partition(i, **kw) for i in range(n_part)]
    assert sum(p.n_reads for p in parts) == 2000000
    ctx = _lib.Context(0)
    try:
        util.run_gpu(ctx, parts)
        sizes = ctx.sizes()
        assert sizes["n_positions"] > 2e8 and sizes["n_problems"] > 90000
        rep = util.compare_partitions(ctx, parts, (util.run_oracle(p) for p in parts))
        assert rep["y_identical"]
        # the replay (sized arenas, the scoring plan over side streams): results only -- final positions and labels, the whole job's
        first = [a.copy() for a in ctx.download()]             # (the run just checked)
        for _ in range(2):
            ctx.run(); ctx.sync()
            for a, b in zip(first, ctx.download()):
                assert a.shape == b.shape and (a == b).all()
        sy = ctx.tap("sync")
        assert int(sy[6]) == 0                               # no waiter reached its limit (10 ms per 2^18 reads, at most 20)
    finally:
        ctx.close()


def test_a_batch_of_more_than_2_31_positions_is_refused_with_the_word_the_cli_halves_on():
    """fseg_upload takes fewer than 2^31 positions: a batch beyond that is refused before anything is allocated, with the message
    freddie_amd/segment.py:batch_too_large() looks for (the CLI then halves the batch: the reference has no batches to be too large)."""
    import numpy as np
    from freddie_amd import pack, segment
    span = (1 << 30) - 1
    one = pack.pack_partition(np.array([100], np.int32), np.array([100 + span], np.int32), np.array([0, 1], np.int64),
                              np.array([150], np.int32), np.array([900], np.int32), dedupe=True)
    ctx = _lib.Context(0)
    try:
        ctx.set_params(**util.DEFAULTS, **util.param_tables(util.DEFAULTS))
        with pytest.raises(_lib.SegError) as info:
            ctx.upload(**pack.concat_batch([one, one, one]))
        assert segment.batch_too_large(info.value)
        util.run_gpu(ctx, [util.make_partition(3, **{k: v for k, v in synth.WORKLOADS["config4"].items() if k != "n_partitions"})])   # the context is still usable
        assert ctx.sizes()["n_positions"] > 0
    finally:
        ctx.close()
