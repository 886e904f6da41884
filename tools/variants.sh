#!/bin/bash
# compare builds of libfreddie_seg.so under freddie_amd/variants/ (tuning experiments): tools/variants.sh [workloads...]
for w in "${@:-config2 config3}"; do
  for w1 in $w; do
    for so in default freddie_amd/variants/*.so; do
      if [ "$so" = default ]; then unset FSEG_LIB; else export FSEG_LIB=$PWD/$so; fi
      echo "== $so"
      FSEG_NO_GRAPH=1 python bench.py --workload $w1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | python profiles/benchsum.py
    done
  done
done
