"""Unit checks of the oracle's restatements of third-party numerics (scipy / numpy are the
libraries the reference calls; they are installed in the image, the reference is not needed)."""
import numpy as np
import pytest
from scipy.ndimage import gaussian_filter1d
from scipy.signal import find_peaks

from freddie_amd import tables
from oracle import oracle


@pytest.mark.parametrize("sigma,truncate,mode", [(5.0, 4.0, "reflect"), (3.0, 4.0, "reflect"), (2.5, 4.0, "reflect"),
                                                 (5.0, 1.0, "constant"), (3.0, 1.0, "constant"), (12.0, 4.0, "reflect"),
                                                 (50.0, 4.0, "reflect")])
def test_gaussian_bit_exact(sigma, truncate, mode):
    rng = np.random.default_rng(int(sigma * 10) + len(mode))
    w = tables.gaussian_half_kernel(sigma, truncate)
    for n in (2, 5, 37, 500, 3000):
        x = np.zeros(n)
        idx = rng.integers(0, n, max(1, n // 7))
        np.add.at(x, idx, rng.integers(1, 40, len(idx)).astype(float))
        want = gaussian_filter1d(x, sigma, truncate=truncate, mode=mode, cval=0.0)
        got = oracle.gaussian(x, w, mode)
        assert np.array_equal(want, got), (n, np.abs(want - got).max())


def test_numpy_sum_order():
    rng = np.random.default_rng(7)
    for n in (1, 5, 8, 9, 100, 128, 129, 1000, 8191, 8192, 8193, 20000, 100003):
        a = rng.random(n) * rng.integers(1, 1000)
        assert oracle.np_sum(a) == float(np.sum(a)), n
        y = np.where(rng.random(n) < 0.6, a, 0.0)
        v = y[y > 0]
        if len(v):
            assert oracle.variance_threshold(y, 3.0) == float(v.mean() + 3.0 * v.std()), n


def _select_by_distance_stable(peaks, prio, distance):
    """scipy.signal._peak_finding_utils._select_by_peak_distance with a STABLE argsort.  numpy's default
    argsort is an unstable SIMD sort on AVX-512 hosts, so for exactly equal heights closer than
    `distance` the reference's own result depends on the machine; the oracle (and the kernels) fix the
    stable order: among equal heights the later peak is processed first."""
    keep = np.ones(len(peaks), bool)
    order = np.argsort(prio, kind="stable")
    for i in range(len(peaks) - 1, -1, -1):
        j = order[i]
        if not keep[j]:
            continue
        k = j - 1
        while k >= 0 and peaks[j] - peaks[k] < distance:
            keep[k] = False
            k -= 1
        k = j + 1
        while k < len(peaks) and peaks[k] - peaks[j] < distance:
            keep[k] = False
            k += 1
    return peaks[keep]


def test_local_maxima_and_distance():
    rng = np.random.default_rng(3)
    for trial in range(300):
        n = int(rng.integers(3, 200))
        y = rng.integers(0, 4, n).astype(float)          # tie-heavy
        pk = find_peaks(y)[0]
        assert np.array_equal(oracle.local_maxima(y), pk)
        assert np.array_equal(oracle.peaks_with_distance(y, 20), _select_by_distance_stable(pk, y[pk], 20))
        y2 = rng.random(n)                                # no ties: scipy itself is the reference
        assert np.array_equal(oracle.peaks_with_distance(y2, 20), find_peaks(y2, distance=20)[0])
