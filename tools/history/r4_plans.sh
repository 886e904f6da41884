#!/bin/bash
# stage time of one resident batch under several plans, product build:  tools/r4_plans.sh <tag> "<plans>" [workload] [extra env as VAR=VAL ...]
T=${1:-p}; PLANS=$2; WL=${3:-config4}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$T
for sp in ${SPLITS:-7 0}; do
for p in $PLANS; do
  FSEG_SPLIT_DP=$sp FSEG_SCORE_PLAN="$p" timeout -k 10 200 python tools/replay_probe.py --workload $WL 2>&1 | grep replay | cut -c1-110 | sed "s/replay/split=$sp plan $p/"
done
done | tee gpurun_out/$T/plans.txt
