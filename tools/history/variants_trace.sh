#!/bin/bash
# per-kernel medians of the scoring kernels for every build under freddie_amd/variants/: tools/variants_trace.sh [workload] [kernel pattern]
W=${1:-config4}
PAT=${2:-k_solve|k_tiny|k_score|k_dp}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for so in freddie_amd/variants/*.so; do
  n=$(basename $so .so)
  rm -rf gpurun_out/vt_$n
  FSEG_LIB=$PWD/$so FSEG_NO_GRAPH=1 FSEG_NO_FORK=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/vt_$n -o p -- python3 tools/replay_probe.py --workload $W > gpurun_out/vt_$n.txt 2>&1
  echo "== $n: $(grep replay gpurun_out/vt_$n.txt | cut -c1-100)"
  python profiles/trace_medians.py gpurun_out/vt_$n/p_kernel_trace.csv | grep -E "$PAT"
done
