#!/bin/bash
# A/B of two builds of the library on one box: replays of one resident batch (no stage events; one stream and forked), alternating.
#   tools/probes/lib_ab.sh <variant-a> <variant-b> [workload]      (variants: names of tools/build_variant.sh, or "main")
A=$1; B=$2; W=${3:-config4}
lib() { if [ "$1" = main ]; then echo $PWD/freddie_amd/libfreddie_seg.so; else echo $PWD/freddie_amd/libfreddie_seg_$1.so; fi; }
for rep in 1 2 3; do
  for v in $A $B; do
    for e in "" "FSEG_NO_FORK=1"; do
      r=$(env FSEG_LIB=$(lib $v) $e timeout -k 10 120 python tools/replay_probe.py --workload $W --profiling 0 2>/dev/null | grep "^replay" | sed 's/replay: \([0-9.]*\) ms.*/\1/')
      printf "%-8s %-16s %s ms/replay\n" $v "${e:-forked}" "$r"
    done
  done
done
