#!/bin/bash
# Kernel durations (one stream, plain launches, rocprofv3 --kernel-trace --stats) of the product library under environment settings:
#   tools/probes/kernel_env.sh "<pattern> ..." "VAR=a" "VAR=b" ...
PATS=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
k=0
for e in "$@"; do
  k=$((k+1)); rm -rf gpurun_out/ke_$k
  env $e FSEG_NO_FORK=1 FSEG_NO_GRAPH=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ke_$k -o kt -- python3 tools/replay_probe.py --workload ${WORKLOAD:-config4} --profiling 0 > gpurun_out/ke_$k.log 2>&1 || exit 1
  python3 - "$e" gpurun_out/ke_$k $PATS <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[2] + "/**/*kernel_stats.csv", recursive=True)
print("==", sys.argv[1])
for r in csv.DictReader(open(f[0])):
    if any(k in r["Name"] for k in sys.argv[3:]):
        print("  %-60s calls %5s avg %8.2f us  min %7.2f  max %7.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
