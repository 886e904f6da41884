#!/bin/bash
# stage bracket of a resident batch under library builds side by side:  tools/r4_libs.sh <tag> <workload> "<variants>"   (main = the tree's build, x = libfreddie_seg_x.so)
T=$1; WL=$2; VARS=$3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$T
for rep in 1 2 3; do
for v in $VARS; do
  if [ "$v" = main ]; then unset FSEG_LIB; else export FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_$v.so; fi
  timeout -k 10 200 python tools/replay_probe.py --workload $WL 2>&1 | grep replay | cut -c1-105 | sed "s|replay|$v|"
done
done | tee -a gpurun_out/$T/libs.txt
