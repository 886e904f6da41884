#!/bin/bash
# Quick GPU check of a scoring-stage change: parity tests, per-problem clocks, one-context kernel trace, short bench.
#   tools/r3_quick.sh <tag> [tests...]
T=${1:-q}; shift
O=gpurun_out/$T
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
TESTS=${@:-tests/test_gpu_paths.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py}
timeout -k 10 900 python -m pytest $TESTS -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -5 $O/tests.txt
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
if [ -f freddie_amd/libfreddie_seg_timing.so ]; then
  FSEG_NO_FORK=1 FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_timing.so timeout -k 10 300 python tools/prob_ticks.py config4 > $O/prob_ticks_nofork.txt 2>&1
  FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_timing.so timeout -k 10 300 python tools/prob_ticks.py config4 > $O/prob_ticks.txt 2>&1
  cat $O/prob_ticks.txt | grep -v slowest
fi
FSEG_NO_FORK=1 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -o p -- python3 bench.py --contexts 1 --no-cpu-baseline --no-e2e --no-extras --steps 16 > $O/bench_c1.json 2> $O/trace1.err
python profiles/trace_medians.py $O/trace1/p_kernel_trace.csv > $O/config4_kernel_medians.txt; head -30 $O/config4_kernel_medians.txt
timeout -k 10 600 python bench.py --no-cpu-baseline --no-e2e > $O/bench.json 2> $O/bench.err; python profiles/benchsum.py < $O/bench.json
rm -rf $O/trace1
