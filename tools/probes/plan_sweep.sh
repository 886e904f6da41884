#!/bin/bash
# The scoring stage alone under other stream plans (FSEG_SCORE_PLAN): tools/probes/plan_sweep.sh [workloads...]
W=${@:-config4 config3 config5}
for w in $W; do
  echo "== $w"
  for p in "gM|W|hB|gST" "gM|W|hBT|gS" "gM|W|hB|gTS" "gMT|W|hB|gS" "gM|W|hBS|gT" "gM|W|hB|gS|gT" "gMS|W|hB|gT" "gM|W|hBgT|gS" "gM|WhB|gST" "gM|W|hB|ST"; do
    r=$(FSEG_SCORE_PLAN="$p" python tools/replay_probe.py --workload $w --profiling 2 2>/dev/null | grep "^replay" | sed 's/.*scoring \([0-9.]*\) ms.*/\1/')
    printf "  %-18s %s\n" "$p" "$r"
  done
done
