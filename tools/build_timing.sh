#!/bin/bash
# Diagnostic build with phase / per-problem clocks (tools/prob_ticks.py, tools/solve_timing.py); never used for reported numbers.
cd "$(dirname "$0")/.." && python -m freddie_amd.build --variant timing -DFSEG_SCORE_TIMING "$@"
