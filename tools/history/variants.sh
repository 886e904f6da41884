#!/bin/bash
# compare builds of libfreddie_seg.so under freddie_amd/variants/ (tuning experiments): tools/variants.sh [workload] [reps]
# (FSEG_LIB skips the stale-library check: variants are built by hand with extra -D flags)
W=${1:-config4}
for rep in 1 2; do
  for so in freddie_amd/variants/*.so; do
    echo "== $so (run $rep)"
    FSEG_LIB=$PWD/$so FSEG_NO_GRAPH=1 python tools/replay_probe.py --workload $W 2>&1 | tail -2
  done
done
