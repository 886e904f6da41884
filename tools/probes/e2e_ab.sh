#!/bin/bash
# A/B of the drop-in CLI on one box: tools/probes/e2e_ab.sh <other_tree> [repeats]   (the other tree: a checkout with its libraries built)
OTHER=$1; N=${2:-4}
D=/dev/shm/e2e_ab
python tools/e2e_bench.py --partitions 4000 --reads 500 --generate-only --keep $D > /dev/null
run() { # tree
  rm -rf $D/out; local t0=$(date +%s.%N)
  python $1/py/freddie_segment.py -s $D/split -o $D/out -t 16 --gpus 1 --sidecar off > /dev/null
  local t1=$(date +%s.%N); python3 -c "print('%.3f' % ($t1 - $t0))"
}
run . > /dev/null; run $OTHER > /dev/null
for i in $(seq $N); do echo "new $(run .)   old $(run $OTHER)"; done
rm -rf $D
