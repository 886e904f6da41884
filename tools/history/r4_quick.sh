#!/bin/bash
# Round-4 GPU check of a scoring-stage change: DP probe, parity tests, the stage against other builds and under other plans.
#   tools/r4_quick.sh <tag> "<variants>" "<plans>" [tests...]
T=${1:-q}; VARS=${2:-main}; PLANS=${3:-}; shift; shift; shift
O=gpurun_out/$T
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
if [ -x tools/probes/dp_probe.bin ] && [ -z "$NO_PROBE" ]; then timeout -k 10 300 tools/probes/dp_probe.bin > $O/dp_probe.txt 2>&1; grep -c MISMATCH $O/dp_probe.txt; grep "grid=  1" $O/dp_probe.txt | head -40; fi
TESTS=${@:-tests}
if [ -z "$NO_TESTS" ]; then
timeout -k 10 1100 python -m pytest $TESTS -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -5 $O/tests.txt
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
fi
for v in $VARS; do
  if [ "$v" = main ]; then unset FSEG_LIB; else export FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_$v.so; fi
  for wl in ${WLS:-config4}; do
  echo "== $v $wl"
  timeout -k 10 200 python tools/replay_probe.py --workload $wl 2>&1 | grep replay
  FSEG_NO_FORK=1 timeout -k 10 200 python tools/replay_probe.py --workload $wl 2>&1 | grep replay | sed 's/replay/replay (one stream)/'
  for p in $PLANS; do
    FSEG_SCORE_PLAN="$p" timeout -k 10 200 python tools/replay_probe.py --workload $wl 2>&1 | grep replay | sed "s/replay/replay (plan $p)/"
  done
  done
  FSEG_NO_FORK=1 FSEG_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -o p -- python3 tools/replay_probe.py --workload config4 > /dev/null 2> $O/trace_$v.err
  python profiles/trace_medians.py $O/trace_$v/p_kernel_trace.csv > $O/medians_$v.txt
  grep -E "k_wave|k_solve|k_tiny|k_score|k_dp|k_cov" $O/medians_$v.txt
  rm -rf $O/trace_$v
done
unset FSEG_LIB
if [ -f freddie_amd/libfreddie_seg_timing.so ] && [ -z "$NO_TICKS" ]; then
  FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_timing.so timeout -k 10 300 python tools/prob_ticks.py config4 > $O/prob_ticks.txt 2>&1
  grep -v slowest $O/prob_ticks.txt | head -30
  PROB_TICKS_PARTS=12 FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_timing.so timeout -k 10 300 python tools/prob_ticks.py config4 > $O/prob_ticks_alone.txt 2>&1
  grep -v slowest $O/prob_ticks_alone.txt | head -30
fi
