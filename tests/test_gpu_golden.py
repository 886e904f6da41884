"""GPU parity against the golden fixtures: the HIP path vs outputs of the reference itself."""
import pytest

import goldens
import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", goldens.names())
def test_hip_matches_reference_golden(gpu_ctx, name):
    g = goldens.load(name)
    part = goldens.partition_of(g)
    util.run_gpu(gpu_ctx, [part], goldens.params_of(g), goldens.tables_of(g))
    rep = util.compare_partitions(gpu_ctx, [part], [goldens.as_oracle_result(g)])   # Y within 1e-6 (north_star)
    assert rep["y_identical"], "smoothed signal not bit-identical (max err %g)" % rep["max_y_err"]


@pytest.mark.parametrize("name", goldens.raising_names())
def test_hip_refuses_what_the_reference_raises_on(gpu_ctx, name):
    """x_*.npz: the reference raises in break_large_problems (:640); the library must refuse the batch, not return labels."""
    from freddie_amd import _lib
    g = goldens.load(name)
    with pytest.raises(_lib.SegError):
        util.run_gpu(gpu_ctx, [goldens.partition_of(g)], goldens.params_of(g), goldens.tables_of(g))
        gpu_ctx.download()
