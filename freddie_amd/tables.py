"""Run-time tables the reference obtains from numpy / scipy / Python and that the C-ABI takes as data.

They are computed here the way the reference's dependencies compute them, so that with the
same numpy the kernels see bit-identical constants:

* ``gaussian_half_kernel`` -- scipy.ndimage._filters._gaussian_kernel1d as used by
  ``gaussian_filter1d(y, sigma, truncate=4.0)`` (py/freddie_segment.py:755) and by
  ``gaussian_filter1d(..., mode='constant', truncate=1.0)`` (:260-261).  numpy's vectorised
  ``exp`` and its pairwise ``sum`` are part of the result, hence numpy is used, not libm.
* ``smooth_threshold`` -- py/freddie_segment.py:277-286 (Python ``round(y, 2)``).
"""
import math

import numpy as np


def gaussian_radius(sigma, truncate):
    return int(truncate * float(sigma) + 0.5)


def gaussian_half_kernel(sigma, truncate):
    radius = gaussian_radius(sigma, truncate)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    return np.ascontiguousarray(phi[radius:], dtype=np.float64)


def smooth_threshold(threshold):
    table = []
    while True:
        seg_len = len(table)
        y = threshold / (1 + ((threshold - .5) / .5) * math.exp(-0.05 * seg_len))
        if seg_len > 5 and seg_len * (threshold - y) < 0.5:
            return table
        table.append(round(y, 2))
        assert len(table) < 1000
