#!/bin/bash
# The benchmark's headline for another workload under several switches: tools/bench_wl.sh <tag> <workload> <steps> "<ENV=.. ENV=..>" ...
T=$1; WL=$2; ST=$3; shift 3
O=gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "== $WL $cfg"
  env $cfg timeout -k 10 500 python bench.py --workload $WL --no-cpu-baseline --no-e2e --steps $ST > $O/bench_${WL}_$i.json 2> $O/bench_${WL}_$i.err; python profiles/benchsum.py < $O/bench_${WL}_$i.json | head -1
done
