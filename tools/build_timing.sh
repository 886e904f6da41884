#!/bin/bash
# Diagnostic build with phase / per-problem clocks (tools/prob_ticks.py, tools/solve_timing.py); never used for reported numbers.
cd "$(dirname "$0")/.." && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC -I include -DFSEG_SCORE_TIMING "$@" \
    -o freddie_amd/libfreddie_seg_timing.so freddie_amd/csrc/freddie_seg.hip freddie_amd/csrc/freddie_seg_sort.hip -lhsa-runtime64
