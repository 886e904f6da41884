#!/bin/bash
# kernel gaps of an unprofiled single-graph replay (the product's replay path):  tools/r4_graphgaps.sh <tag> [workload]
T=$1; WL=${2:-config4}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$T
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$T/trace -o p -- python3 tools/replay_probe.py --workload $WL --profiling 0 > gpurun_out/$T/replay.txt 2> gpurun_out/$T/trace.err
python tools/run_gaps.py gpurun_out/$T/trace/p_kernel_trace.csv | tee gpurun_out/$T/gaps.txt
python tools/stage_span.py gpurun_out/$T/trace/p_kernel_trace.csv | tee gpurun_out/$T/span.txt
tail -2 gpurun_out/$T/replay.txt
