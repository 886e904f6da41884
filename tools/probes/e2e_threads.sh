#!/bin/bash
# The drop-in CLI's wall time against -t on one box (cpu.max of the lease: 16 CPUs' worth per 100 ms; every pool of the pipeline takes -t threads;
# (n) behind a wall time: the 100 ms periods in which the cgroup was throttled).  Further arguments: VAR=value settings to repeat the sweep under.
#   tools/probes/e2e_threads.sh "4 6 8 12 16" [repeats]
TS=${1:-"4 6 8 12 16"}; N=${2:-3}
D=/dev/shm/e2e_t
python tools/e2e_bench.py --partitions 4000 --reads 500 --generate-only --keep $D > /dev/null
cat /sys/fs/cgroup/cpu.max 2>/dev/null
thr() { awk '/nr_throttled/{print $2}' /sys/fs/cgroup/cpu.stat 2>/dev/null; }
run() { rm -rf $D/out; local n0=$(thr); local t0=$(date +%s.%N)
  env $2 python py/freddie_segment.py -s $D/split -o $D/out -t $1 --gpus 1 --sidecar off > /dev/null
  local t1=$(date +%s.%N); local n1=$(thr); python3 -c "print('%.3f(%d)' % ($t1 - $t0, ${n1:-0} - ${n0:-0}), end=' ')"; }   # (n): periods in which the cgroup was throttled
run 16 > /dev/null
for t in $TS; do echo -n "-t $t: "; for i in $(seq $N); do run $t; done; echo; done
shift 2; for e in "$@"; do for t in $TS; do echo -n "-t $t $e: "; for i in $(seq $N); do run $t "$e"; done; echo; done; done
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null
rm -rf $D
