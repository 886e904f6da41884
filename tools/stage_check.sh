#!/bin/bash
# Round-5 GPU check of a scoring-stage change: parity tests of the stage's paths, then the stage's bracket (HIP events, one
# resident batch replayed) for the three many-partition workloads with the device-side fork / join and with events, and the
# stage's kernels one by one from a kernel trace.
#   tools/stage_check.sh <tag> [tests...]          (NO_TESTS=1 skips the tests; ENVS="A=1;B=2 C=3" adds environment variants; NO_TRACE=1)
T=${1:-r5s}; shift
O=gpurun_out/$T
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
TESTS=${@:-tests/test_gpu_paths.py tests/test_gpu_parity.py}
if [ -z "$NO_TESTS" ]; then
  timeout -k 10 ${TEST_TMO:-400} python -m pytest $TESTS -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -5 $O/tests.txt
  if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
fi
IFS=';' read -ra EV <<< "-;FSEG_DEV_SYNC=0${ENVS:+;$ENVS}"
for wl in ${WLS:-config4 config5 config3}; do
  for e in "${EV[@]}"; do
    if [ "$e" = "-" ]; then v=""; else v="$e"; fi
    echo "== $wl ${v:-default}"
    env $v timeout -k 10 200 python tools/replay_probe.py --workload $wl --profiling 2 2>&1 | grep -A1 "replay" | cut -c1-230
  done
done 2>&1 | tee $O/stage_ms.txt
# TRACE1=1: per-kernel medians of the one-stream replay (what every kernel takes with the GPU to itself)
if [ -n "$TRACE1" ]; then
  for wl in ${TRACE1_WLS:-config4}; do
    FSEG_NO_FORK=1 FSEG_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace1_$wl -o p -- python3 tools/replay_probe.py --workload $wl > /dev/null 2> $O/trace1_$wl.err
    python profiles/trace_medians.py $O/trace1_$wl/p_kernel_trace.csv > $O/${wl}_kernel_medians.txt; cat $O/${wl}_kernel_medians.txt
    rm -rf $O/trace1_$wl
  done
fi
if [ -n "$NO_TRACE" ]; then exit 0; fi
for wl in ${TRACE_WLS:-config4 config3}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$wl -o p -- python3 tools/replay_probe.py --workload $wl --profiling 2 > $O/replay_$wl.txt 2> $O/trace_$wl.err
  python tools/stage_timeline.py $O/trace_$wl/p_kernel_trace.csv > $O/${wl}_stage_timeline.txt
  python tools/stage_span.py $O/trace_$wl/p_kernel_trace.csv > $O/${wl}_stage_span.txt; grep replay $O/replay_$wl.txt | cut -c1-120 >> $O/${wl}_stage_span.txt
  cat $O/${wl}_stage_timeline.txt $O/${wl}_stage_span.txt
  rm -rf $O/trace_$wl
done
