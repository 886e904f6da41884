#!/bin/bash
# per-stage times of one resident batch (plain launches, events around every stage) under environment variants:
#   tools/r4_env.sh <tag> [workload] -- "<VAR=VAL ...>" ...     ("-" = default)
T=$1; WL=${2:-config4}; shift; shift; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$T
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  echo "== $v"
  env $e FSEG_NO_GRAPH=1 timeout -k 10 200 python tools/replay_probe.py --workload $WL 2>&1 | tail -1
done | tee gpurun_out/$T/env.txt
