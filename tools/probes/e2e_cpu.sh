#!/bin/bash
# CPU seconds (user, system) and throttled periods of one run of the drop-in CLI: is the 16-CPU quota what its wall time runs out of?
D=/dev/shm/e2e_c
python tools/e2e_bench.py --partitions 4000 --reads 500 --generate-only --keep $D > /dev/null
for i in 1 2 3 4; do
  rm -rf $D/out
  n0=$(awk '/nr_throttled/{print $2}' /sys/fs/cgroup/cpu.stat); u0=$(awk '/throttled_usec/{print $2}' /sys/fs/cgroup/cpu.stat); c0=$(awk '/^usage_usec/{print $2}' /sys/fs/cgroup/cpu.stat)
  env $1 python3 - $D <<'PY'
import os, resource, subprocess, sys, time
D = sys.argv[1]
t0 = time.perf_counter()
subprocess.run([sys.executable, "py/freddie_segment.py", "-s", D + "/split", "-o", D + "/out", "-t", "16", "--gpus", "1", "--sidecar", "off"], check=True, stdout=subprocess.DEVNULL)
w = time.perf_counter() - t0
r = resource.getrusage(resource.RUSAGE_CHILDREN)
print("wall %.3f s, user %.2f s, sys %.2f s, max RSS %d MB, minor faults %d, vol ctx %d, invol ctx %d" % (w, r.ru_utime, r.ru_stime, r.ru_maxrss // 1024, r.ru_minflt, r.ru_nvcsw, r.ru_nivcsw))
PY
  n1=$(awk '/nr_throttled/{print $2}' /sys/fs/cgroup/cpu.stat); u1=$(awk '/throttled_usec/{print $2}' /sys/fs/cgroup/cpu.stat); c1=$(awk '/^usage_usec/{print $2}' /sys/fs/cgroup/cpu.stat)
  echo "   cgroup: usage $(( (c1 - c0) / 1000 )) ms, throttled periods $((n1 - n0)), throttled thread-time $(( (u1 - u0) / 1000 )) ms"
done
rm -rf $D
