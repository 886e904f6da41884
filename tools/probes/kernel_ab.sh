#!/bin/bash
# Kernel durations of two builds of the library, one stream, plain launches (rocprofv3 --kernel-trace --stats):
#   tools/probes/kernel_ab.sh <variant-a> <variant-b> [pattern ...]
A=$1; B=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in $A $B; do
  L=$PWD/freddie_amd/libfreddie_seg_$v.so; [ $v = main ] && L=$PWD/freddie_amd/libfreddie_seg.so
  rm -rf gpurun_out/kt_$v
  FSEG_LIB=$L FSEG_NO_FORK=1 FSEG_NO_GRAPH=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$v -o kt -- python3 tools/replay_probe.py --workload config4 --profiling 0 > gpurun_out/kt_$v.log 2>&1 || exit 1
done
python3 - "$A" "$B" "$@" <<'PY'
import csv, glob, sys
pats = sys.argv[3:] or ["k_"]
for v in sys.argv[1:3]:
    f = glob.glob("gpurun_out/kt_%s/**/*kernel_stats.csv" % v, recursive=True)
    print("==", v)
    tot = 0.0
    for r in csv.DictReader(open(f[0])):
        n = r["Name"]
        tot += float(r["TotalDurationNs"])
        if any(k in n for k in pats):
            print("  %-46s calls %5s avg %8.2f us  min %7.2f  max %7.2f" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    print("  sum of all kernels' time: %.1f us" % (tot / 1e3))
PY
