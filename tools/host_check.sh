#!/bin/bash
# Round-5 host-side scaling check on the GPU box (no GPU needed: tools/host_ceiling.py replaces the device by a stand-in):
# the 2 M-read job and a job four times as large, partitions spread over 24 contig directories, 1 .. 8 workers.
T=${1:-r5host}; O=gpurun_out/$T
cd $GRAFT_REPO_ROOT; mkdir -p $O
df -h /dev/shm | tail -1; nproc; free -g | head -2
for job in "4000 2m" "${BIG:-16000} big"; do
  set -- $job
  FREDDIE_TIMING=1 timeout -k 10 ${TMO:-400} python tools/host_ceiling.py --partitions $1 --reads 500 --contigs 24 --workers ${WORKERS:-1,2,4,8} --label ${LABELS:-0} --repeat 2 > $O/host_ceiling_$2.txt 2> $O/host_ceiling_$2.err
  cat $O/host_ceiling_$2.txt; grep "workers started" $O/host_ceiling_$2.err | cut -c1-260
done
