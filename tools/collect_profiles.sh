#!/bin/bash
# copy the summaries of a tools/profile_round.sh run (gpurun_out/<tag>) into profiles/ under the round prefix:
#   tools/collect_profiles.sh <tag> <prefix>        e.g. tools/collect_profiles.sh r02b r02
T=$1; P=$2; O=gpurun_out/$T
set -e
cp $O/bench_default.json profiles/${P}_bench_default.json
cp $O/bench_under_rocprof.json profiles/${P}_bench_under_rocprof.json
cp $O/stats/p_kernel_stats.csv profiles/${P}_config4_kernel_stats.csv
cp $O/config4_kernel_medians.txt profiles/${P}_config4_kernel_medians.txt
cp $O/config4_kernel_medians_3ctx.txt profiles/${P}_config4_kernel_medians_3ctx.txt
for w in config4 config2; do
  cp $O/pmc_${w}_FETCH_SIZE/p_counter_collection.csv profiles/${P}_${w}_pmc_fetch_size.csv
  cp $O/pmc_${w}_WRITE_SIZE/p_counter_collection.csv profiles/${P}_${w}_pmc_write_size.csv
  cp $O/pmc_${w}_sq/p_counter_collection.csv profiles/${P}_${w}_pmc_sq.csv
  cp $O/${w}_pmc_summary.txt profiles/${P}_${w}_pmc_summary.txt
  cp $O/${w}_sq_summary.txt profiles/${P}_${w}_sq_summary.txt
done
for w in config2 config3 config5; do cp $O/${w}_bench.json profiles/${P}_${w}_bench.json; done
cp $O/n3_cluster_many_bench.json profiles/${P}_n3_cluster_many_bench.json
cp $O/n3_cluster_big_bench.json profiles/${P}_n3_cluster_big_bench.json
cp $O/n4_isoforms_bench.json profiles/${P}_n4_isoforms_bench.json
cp $O/e2e.log profiles/${P}_e2e.log
python profiles/make_traffic.py $P
