# k_label_reads's average duration under rocprofv3 for variant builds of the library (diagnostic switches: wrong results, timing only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  L=$PWD/freddie_amd/libfreddie_seg_$v.so; [ "$v" = product ] && L=$PWD/freddie_amd/libfreddie_seg.so
  FSEG_LIB=$L FSEG_NO_FORK=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sm_$v -o p -- python3 tools/replay_probe.py --workload config4 > gpurun_out/sm_$v.log 2>&1
  python3 - <<P
import csv
for r in csv.DictReader(open('gpurun_out/sm_$v/p_kernel_stats.csv')):
    if 'k_label_reads' in r['Name']: print('$v', 'k_label_reads', r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us')
P
done
