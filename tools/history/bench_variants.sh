#!/bin/bash
# The benchmark's headline (8 contexts, host to host) for several builds / switches: tools/bench_variants.sh <tag> "<ENV=.. ENV=..>" ...
T=$1; shift
O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "== $cfg"
  env $cfg timeout -k 10 400 python bench.py --no-cpu-baseline --no-e2e --steps 192 > $O/bench_$i.json 2> $O/bench_$i.err; python profiles/benchsum.py < $O/bench_$i.json | head -1
done
