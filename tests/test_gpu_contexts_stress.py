"""Contexts of one device under random interleavings (``-m gpu``): what the pool map of the reference (py/freddie_segment.py:871-876)
becomes here is several contexts taking turns on a GPU, and since round 6 only the context that has CLAIMED the device forks over side
streams (device-side waiters only on side streams a probe has seen run beside the main stream).  Every thread drives one context through
a random sequence of new batches, replays of the resident batch, result fetches in both label forms and -- now and then -- a context
destroyed and made again while the others are in flight; every result is compared with the oracle, and no waiter may reach its limit."""
import os
import random
import threading

import numpy as np
import pytest

import util
from freddie_amd import _lib, synth

pytestmark = pytest.mark.gpu

# FREDDIE_STRESS_SEEDS=1,2,3,...: a longer campaign by hand
SEEDS = [int(x) for x in os.environ.get("FREDDIE_STRESS_SEEDS", "1,2").split(",")]


def _batch_pool():
    kw = dict(synth.WORKLOADS["config4"]); kw.pop("n_partitions")
    pool = []
    for b in range(6):
        n = (70, 48, 12, 90, 1, 66)[b]                                     # plans need many problems of every class; small batches keep to one stream
        parts = [util.make_partition(88000 + 200 * b + i, **kw) for i in range(n)]
        pool.append((parts, [util.run_oracle(p) for p in parts]))
    return pool


@pytest.fixture(scope="module")
def pool():
    return _batch_pool()


@pytest.mark.parametrize("seed", SEEDS)
def test_random_interleavings_of_contexts(seed, pool):
    rng = random.Random(seed)
    n_ctx = rng.choice([2, 3, 5, 8])
    plans = [[rng.choice(["new", "new", "replay", "replay", "fetch", "recreate"]) for _ in range(14)] for _ in range(n_ctx)]
    picks = [[rng.randrange(len(pool)) for _ in range(14)] for _ in range(n_ctx)]
    reps = [[rng.randrange(1, 6) for _ in range(14)] for _ in range(n_ctx)]
    errors, timeouts, forked = [], [0] * n_ctx, [0] * n_ctx
    start = threading.Barrier(n_ctx)

    def worker(k):
        ctx = _lib.Context(0)
        cur = None
        try:
            start.wait()
            for step, what in enumerate(plans[k]):
                if what == "new" or cur is None:
                    cur = pool[picks[k][step]]
                    util.run_gpu(ctx, cur[0])
                    assert util.compare_partitions(ctx, *cur)["y_identical"]
                elif what == "replay":
                    for _ in range(reps[k][step]):
                        ctx.run(); ctx.sync()
                    assert util.compare_partitions(ctx, *cur)["y_identical"]
                elif what == "fetch":
                    ctx.run()
                    packed = ctx.results(packed=True)                       # (results of a pending run: the fetch waits for it)
                    plain = ctx.download()
                    assert np.array_equal(packed[3], util.pack_labels(plain[3]))
                    want = np.concatenate([(o["labels"] + 48).astype(np.uint8).ravel() for o in cur[1]])
                    assert np.array_equal(plain[3], want)
                else:                                                       # a context leaves and another comes while the rest run
                    sy = ctx.tap("sync") if cur is not None else None
                    if sy is not None:
                        timeouts[k] += int(sy[6]); forked[k] += int(sy[7])
                    ctx.close()
                    ctx = _lib.Context(0)
                    cur = None
            if cur is not None:
                sy = ctx.tap("sync")
                timeouts[k] += int(sy[6]); forked[k] += int(sy[7])
        except BaseException as exc:                                        # noqa: BLE001
            errors.append(exc)
            start.abort()
        finally:
            ctx.close()

    th = [threading.Thread(target=worker, args=(k,)) for k in range(n_ctx)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise errors[0]
    assert sum(timeouts) == 0, "waiter time-outs per context: %r (forked runs %r)" % (timeouts, forked)
