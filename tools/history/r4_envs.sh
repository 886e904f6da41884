#!/bin/bash
# stage bracket of a resident batch under environment variants:  tools/r4_envs.sh <tag> <workload> "<VAR=VAL ...>" ...   ("-" = default)
T=$1; WL=$2; shift; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$T
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e timeout -k 10 200 python tools/replay_probe.py --workload $WL 2>&1 | grep replay | cut -c1-110 | sed "s/replay/$v/"
done
done | tee gpurun_out/$T/envs.txt
