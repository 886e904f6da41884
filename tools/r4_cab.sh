cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b5
for v in cab1 cab2 cab3; do
  echo "== $v"
  FSEG_LIB=$PWD/freddie_amd/libfreddie_seg_$v.so FSEG_NO_FORK=1 FSEG_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b5/trace_$v -o p -- python3 tools/replay_probe.py --workload config4 > /dev/null 2> gpurun_out/b5/trace_$v.err
  python profiles/trace_medians.py gpurun_out/b5/trace_$v/p_kernel_trace.csv | grep -E "k_cover"
  rm -rf gpurun_out/b5/trace_$v
done
