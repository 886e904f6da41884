#!/bin/bash
# the CLI on a 2 M-read split directory in tmpfs, with the library's trace and the CLI's timing:  tools/e2e_timing.sh <tag> [runs]
T=$1; N=${2:-3}; shift; shift; EXTRA="$@"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$T
python - <<'P' > gpurun_out/$T/gen.txt 2>&1
import os, sys, time, shutil
sys.path.insert(0, os.getcwd())
import bench
import multiprocessing as mp
w = dict(bench.synth.WORKLOADS["config4"]); w.pop("n_partitions")
work = "/dev/shm/freddie_e2e_probe"
if os.path.isdir(os.path.join(work, "split")): sys.exit(0)
split = os.path.join(work, "split")
with mp.get_context("fork").Pool(16) as pool:
    pool.map(bench._gen_split, [(i, w, split) for i in range(4000)], chunksize=8)
print("generated")
P
for i in $(seq $N); do
  rm -rf /dev/shm/freddie_e2e_probe/out
  s=$(date +%s%N)
  FREDDIE_TIMING=1 python py/freddie_segment.py -s /dev/shm/freddie_e2e_probe/split -o /dev/shm/freddie_e2e_probe/out -t 16 --gpus 1 --sidecar off $EXTRA > /dev/null 2> gpurun_out/$T/run_$i.err
  e=$(date +%s%N)
  echo "run $i wall $(( (e - s) / 1000000 )) ms  $EXTRA"
  grep "discover" gpurun_out/$T/run_$i.err | cut -c1-230
done
