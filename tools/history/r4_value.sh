#!/bin/bash
# the job's host-to-host value under environment variants:  tools/r4_value.sh <tag> "<VAR=VAL ...>" ...   (each argument one variant; "-" = default)
T=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$T
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e timeout -k 10 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-28s value %.1f M reads/s  job %.3f ms  stage alone %.3f ms  concurrent %.3f ms' % ('$v', d['value']/1e6, d['ms_per_step'], d['roofline']['launch_ms'], d['roofline_concurrent']['launch_ms']))"
done
done | tee gpurun_out/$T/value.txt
