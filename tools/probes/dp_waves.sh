#!/bin/bash
# The scoring stage alone (one context, one resident batch, events around the stage) with the large class's split DPs by eight waves (default)
# or by one (FSEG_SPLIT_DP=15).  (Round 6: the build this was written for had FSEG_DP_WAVES="s,m,l" -- 1, 2, 4 or 8 waves per class; what it
# measured is in seg_score_fused.hip's comment on k_dpw and in HISTORY.md, Appendix D.)
#   tools/probes/dp_waves.sh "config4 config3 config5"
for w in ${1:-config4 config3 config5}; do
  echo "== $w"
  for rep in 1 2; do
  for e in 7 15; do
    r=$(FSEG_SPLIT_DP=$e timeout -k 10 120 python tools/replay_probe.py --workload $w --profiling 2 2>/dev/null | grep "^replay" | sed 's/.*scoring \([0-9.]*\) ms.*/\1/')
    printf "  FSEG_SPLIT_DP=%-3s %s\n" $e "$r"
  done; done
done
