"""Arena sizing: a fresh context starts with small arenas; the first run of a batch that needs far more
(many partitions, thousands of problems and work items) must size itself and still be exact."""
import numpy as np
import pytest

import util
from freddie_amd import _lib

pytestmark = pytest.mark.gpu


def test_fresh_context_grows_arenas_and_matches_oracle():
    parts = [util.make_partition(200 + i, n_reads=300, n_exons=120, rp=0.05) for i in range(120)]
    oracles = [util.run_oracle(p) for p in parts]
    ctx = _lib.Context(0)
    try:
        util.run_gpu(ctx, parts)
        assert ctx.sizes()["n_problems"] > 1024          # more than the initial problem / work capacity
        util.compare_partitions(ctx, parts, oracles)
        small = [util.make_partition(7, n_reads=50, n_exons=10)]
        util.run_gpu(ctx, small)                          # shrinking batch on the same context
        util.compare_partitions(ctx, small, [util.run_oracle(small[0])])
    finally:
        ctx.close()
