#!/bin/bash
# Round profile: tools/profile_round.sh <tag>   (run on the GPU box from the repo root; writes under gpurun_out/<tag>/)
T=${1:-round}
O=gpurun_out/$T
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python bench.py > $O/config2_bench.json 2> $O/bench.err; python profiles/benchsum.py < $O/config2_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 bench.py > $O/config2_bench_under_rocprof.json 2> $O/stats.err
python profiles/benchsum.py < $O/config2_bench_under_rocprof.json | head -1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 bench.py --no-cpu-baseline --steps 10 > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 bench.py --no-cpu-baseline --steps 10 > /dev/null 2> $O/write.err
python profiles/pmc_summary.py $O/fetch/p_counter_collection.csv $O/write/p_counter_collection.csv > $O/pmc_summary.txt; head -6 $O/pmc_summary.txt
for w in config3 config5; do python bench.py --workload $w --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null > $O/${w}_bench.json; python profiles/benchsum.py < $O/${w}_bench.json | head -1; done
python bench.py --workload config4 --steps 40 --warmup 3 --cpu-all-cores 2>/dev/null > $O/config4_bench.json; python profiles/benchsum.py < $O/config4_bench.json | head -1
FSEG_NO_GRAPH=1 FSEG_NO_FORK=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace_config4 -o p -- python3 bench.py --workload config4 --steps 10 --no-cpu-baseline > /dev/null 2> $O/trace4.err
python profiles/trace_medians.py $O/trace_config4/p_kernel_trace.csv > $O/config4_kernel_medians.txt
FSEG_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace_config2 -o p -- python3 bench.py --steps 10 --no-cpu-baseline > /dev/null 2> $O/trace2.err
python profiles/trace_medians.py $O/trace_config2/p_kernel_trace.csv > $O/config2_kernel_medians.txt
python bench.py --workload cluster-many > $O/n3_cluster_many_bench.json 2>/dev/null
python bench.py --workload cluster-big --steps 2 --no-cpu-baseline > $O/n3_cluster_big_bench.json 2>/dev/null
python bench.py --workload isoforms > $O/n4_isoforms_bench.json 2>/dev/null
python tools/e2e_bench.py --partitions 2000 --reads 500 --threads 8 --sidecar off --repeat 2 > $O/e2e.log 2>&1
python tools/e2e_bench.py --partitions 2000 --reads 500 --threads 8 --sidecar write --repeat 3 >> $O/e2e.log 2>&1
tail -5 $O/e2e.log
