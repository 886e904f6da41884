#!/bin/bash
# copy the summaries of a tools/profile_round.sh run (gpurun_out/<tag>) into profiles/ under the round prefix (no raw counter CSVs):
#   tools/collect_profiles.sh <tag> <prefix>        e.g. tools/collect_profiles.sh r03a r03
T=$1; P=$2; O=gpurun_out/$T
set -e
cp $O/bench_default.json profiles/${P}_bench_default.json
cp $O/bench_under_rocprof.json profiles/${P}_bench_under_rocprof.json
cp $O/stats/p_kernel_stats.csv profiles/${P}_config4_kernel_stats_8ctx.csv
cp $O/trace1/p_kernel_stats.csv profiles/${P}_config4_kernel_stats.csv
cp $O/config4_kernel_medians.txt profiles/${P}_config4_kernel_medians.txt
cp $O/config4_kernel_medians_8ctx.txt profiles/${P}_config4_kernel_medians_8ctx.txt
[ -f $O/config4_waiters_8ctx.txt ] && cp $O/config4_waiters_8ctx.txt profiles/${P}_config4_waiters_8ctx.txt
cp $O/config2_kernel_medians.txt profiles/${P}_config2_kernel_medians.txt
cp $O/config4_stage_span.txt profiles/${P}_config4_stage_span.txt
[ -f $O/config4_stage_timeline.txt ] && cp $O/config4_stage_timeline.txt profiles/${P}_config4_stage_timeline.txt
[ -f $O/config3_stage_timeline.txt ] && cp $O/config3_stage_timeline.txt profiles/${P}_config3_stage_timeline.txt
[ -f $O/config3_stage_span.txt ] && cp $O/config3_stage_span.txt profiles/${P}_config3_stage_span.txt
cp $O/replay_modes.txt profiles/${P}_replay_modes.txt
for w in config4 config2; do
  cp $O/${w}_pmc_summary.txt profiles/${P}_${w}_pmc_summary.txt
  cp $O/${w}_sq_summary.txt profiles/${P}_${w}_sq_summary.txt
done
cp $O/traffic.json profiles/traffic.json
for w in config2 config3 config5; do cp $O/${w}_bench.json profiles/${P}_${w}_bench.json; done
cp $O/n3_cluster_many_bench.json profiles/${P}_n3_cluster_many_bench.json
cp $O/n3_cluster_big_bench.json profiles/${P}_n3_cluster_big_bench.json
cp $O/n4_isoforms_bench.json profiles/${P}_n4_isoforms_bench.json
cp $O/e2e.log profiles/${P}_e2e.log
cp $O/host_ceiling.txt profiles/${P}_host_ceiling.txt
