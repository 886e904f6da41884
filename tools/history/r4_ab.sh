#!/bin/bash
# per-kernel medians (one context, one stream) of builds side by side:  tools/r4_ab.sh <tag> "<variants>" "<kernel regex>"
T=$1; VARS=$2; RE=$3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$T
for v in $VARS; do
  if [ "$v" = main ]; then unset FSEG_LIB; else export FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_$v.so; fi
  echo "== $v"
  FSEG_NO_FORK=1 FSEG_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$T/trace_$v -o p -- python3 tools/replay_probe.py --workload config4 > gpurun_out/$T/replay_$v.txt 2> gpurun_out/$T/trace_$v.err
  python profiles/trace_medians.py gpurun_out/$T/trace_$v/p_kernel_trace.csv | grep -E "$RE"
  tail -1 gpurun_out/$T/replay_$v.txt
  rm -rf gpurun_out/$T/trace_$v
done
