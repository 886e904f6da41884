#!/bin/bash
# Round-5 A/B of the headline figure (the 2 M-read job, host to host, eight contexts) under environment variants, in ONE call
# (box-to-box differences are larger than most changes):  tools/value_ab.sh <tag> "A=1;B=2 C=3"   (REPS=3 rounds of every variant)
T=${1:-r5v}; ENVS=$2
O=gpurun_out/$T
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
IFS=';' read -ra EV <<< "-${ENVS:+;$ENVS}"
for r in $(seq 1 ${REPS:-3}); do
  for e in "${EV[@]}"; do
    if [ "$e" = "-" ]; then v=""; else v="$e"; fi
    env $v timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extras $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-40s %s value %.1f M reads/s  %.3f ms/job  concurrent stage %.3f ms' % ('${v:-default}', '$BENCH_ARGS', d['value']/1e6, d['ms_per_step'], d['roofline_concurrent']['launch_ms']))"
  done
done | tee $O/value_ab.txt
