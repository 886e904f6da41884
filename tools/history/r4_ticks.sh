#!/bin/bash
# per-problem clocks of the scoring stage under several plans (timing build):  tools/r4_ticks.sh <tag> "<plans>" [workload]
T=${1:-t}; PLANS=$2; WL=${3:-config4}
O=gpurun_out/$T
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
i=0
for p in $PLANS; do
  i=$((i+1))
  echo "== plan $p"
  FSEG_SCORE_PLAN="$p" FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_timing.so timeout -k 10 300 python tools/prob_ticks.py $WL > $O/ticks_$i.txt 2>&1
  grep -v "slowest\|fit us" $O/ticks_$i.txt
done
