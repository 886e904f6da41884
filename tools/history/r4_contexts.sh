#!/bin/bash
# the job's host-to-host value by the number of contexts per GPU:  tools/r4_contexts.sh <tag> "<counts>"
T=$1; NS=$2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$T
for rep in 1 2 3; do
for n in $NS; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 40 --warmup 5 --contexts $n --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('contexts %-3s value %.1f M reads/s  job %.3f ms' % ('$n', d['value']/1e6, d['ms_per_step']))"
done
done | tee gpurun_out/$T/contexts.txt
