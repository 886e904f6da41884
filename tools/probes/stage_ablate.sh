#!/bin/bash
# The scoring stage's ablation budget (diagnostic build, wrong results, honest timing): what the stage alone takes when a part of it
# is left out.  tools/build_variant.sh ablate -DFSEG_ABLATE_STAGE=1 first; run from the repo root on the GPU box:
#   tools/probes/stage_ablate.sh [workloads...]  > gpurun_out/stage_ablate.txt
# FSEG_ABLATE bits: 1 the DP launches (k_dpw), 2 the large class, 4 the gates, 8 the mid class, 16 the small class, 32 the tiny class
L=$PWD/freddie_amd/libfreddie_seg_ablate.so
W=${@:-config4 config3 config5}
for w in $W; do
  echo "== $w (stage alone, one resident batch replayed 20 times, events around the stage; ms)"
  for a in 0 1 2 3 4 8 9 16 32 48 58 56 26 42; do
    case $a in 0) n="everything";; 1) n="no DP launches";; 2) n="no large class";; 3) n="no large class, no DP";; 4) n="no gates";; 8) n="no mid class";; 9) n="no mid class, no DP";;
      16) n="no small class";; 32) n="no tiny class";; 48) n="no small, no tiny";; 58) n="only the gates";; 56) n="large class only";; 26) n="tiny class only (+ gates)";; 42) n="small class only (+ gates)";; esac
    r=$(FSEG_LIB=$L FSEG_ABLATE=$a python tools/replay_probe.py --workload $w --profiling 2 2>/dev/null | grep "^replay" | sed 's/.*scoring \([0-9.]*\) ms.*/\1/')
    printf "  FSEG_ABLATE=%-3s %-28s %s\n" $a "$n" "$r"
  done
done
