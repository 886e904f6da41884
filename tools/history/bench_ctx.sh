#!/bin/bash
# The headline for several context counts: tools/bench_ctx.sh <tag> <n> ...
T=$1; shift
O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in "$@"; do
  echo "== contexts $n"
  timeout -k 10 400 python bench.py --contexts $n --no-cpu-baseline --no-e2e --no-extras --steps 192 > $O/bench_ctx$n.json 2> $O/bench_ctx$n.err; python profiles/benchsum.py < $O/bench_ctx$n.json | head -1
done
