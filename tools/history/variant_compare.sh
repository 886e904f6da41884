#!/bin/bash
# Compare builds of the library on one resident config4 batch:  tools/variant_compare.sh <tag> <variant> [<variant> ...]
# ("main" = the product build).  Per variant: scoring stage alone (forked replay and single stream), and the scoring kernels'
# medians from a one-stream kernel trace.
T=$1; shift
O=gpurun_out/$T
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
for v in "$@"; do
  if [ "$v" = main ]; then unset FSEG_LIB; else export FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python tools/replay_probe.py --workload ${WL:-config4} 2>&1 | grep replay
  FSEG_NO_FORK=1 timeout -k 10 200 python tools/replay_probe.py --workload ${WL:-config4} 2>&1 | grep replay | sed 's/replay/replay (one stream)/'
  FSEG_NO_FORK=1 FSEG_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -o p -- python3 tools/replay_probe.py --workload ${WL:-config4} > /dev/null 2> $O/trace_$v.err
  python profiles/trace_medians.py $O/trace_$v/p_kernel_trace.csv > $O/medians_$v.txt
  grep -E "k_wave|k_solve|k_tiny|k_score|k_dp|k_cov|k_lanes" $O/medians_$v.txt
  rm -rf $O/trace_$v
done
