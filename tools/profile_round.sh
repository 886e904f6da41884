#!/bin/bash
# Round profile: tools/profile_round.sh <tag> [prefix]   (run on the GPU box from the repo root; writes under gpurun_out/<tag>/;
# tools/collect_profiles.sh <tag> <prefix> copies the summaries worth keeping into profiles/<prefix>_* afterwards -- never the raw
# counter CSVs; prefix = the round, default the tag's first three characters: it is what profiles/traffic.json cites)
T=${1:-round}
P=${2:-${T:0:3}}
O=gpurun_out/$T
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
# PART=1: bench + traces + counters; PART=2: the other workloads, rows N3 / N4, end to end, the host alone (a call each: gpurun's
# limit is 20 minutes); default: everything
if [[ "${PART:-all}" =~ ^(all|1)$ ]]; then
# 1. the driver's command, plain and under the kernel trace (same command: the averages must agree with the line's events)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; python profiles/benchsum.py < $O/bench_default.json
# (FREDDIE_BENCH_PASSES=1: a step = ONE pass over the job, as until round 5.  With the default 16 passes a step -- 330 k dispatches in
# the process -- rocprofv3 7.2 dies of a SIGSEGV inside its own queue interception a few seconds in, whichever library is loaded,
# round 5's too; one stream per context (FSEG_NO_FORK=1) survives.  Kernel durations do not depend on the number of passes.)
FREDDIE_BENCH_PASSES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $O/bench_under_rocprof.json 2> $O/stats.err
python profiles/benchsum.py < $O/bench_under_rocprof.json | head -1
python profiles/trace_medians.py $O/stats/p_kernel_trace.csv > $O/config4_kernel_medians_8ctx.txt
python tools/waiter_trace.py $O/stats/p_kernel_trace.csv 3 > $O/config4_waiters_8ctx.txt
# 2. one context, one stream, one resident batch replayed: per-kernel durations with the GPU to themselves (the kernels `roofline`
#    times: the split path -- k_solve's rounds + k_dpw -- on one stream)
FSEG_NO_FORK=1 FSEG_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -o p -- python3 tools/replay_probe.py --workload config4 > $O/replay_config4.txt 2> $O/trace1.err
python profiles/trace_medians.py $O/trace1/p_kernel_trace.csv > $O/config4_kernel_medians.txt
FSEG_NO_FORK=1 FSEG_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -o p -- python3 tools/replay_probe.py --workload config2 > $O/replay_config2.txt 2> $O/trace2.err
python profiles/trace_medians.py $O/trace2/p_kernel_trace.csv > $O/config2_kernel_medians.txt
# 2b. the same replay with the DEFAULT plan (side streams, gate): what the stage takes from its first kernel's start to its last
#     kernel's end, and from the end of the kernel before it to the start of the kernel after it (= the event bracket of `roofline`)
#     (--profiling 2: events around the scoring stage only, as in bench.py's roofline leg -- every other stage's pair of events is two more
#     marker packets between the stage's kernels and its neighbours)
rocprofv3 --kernel-trace --output-format csv -d $O/trace3 -o p -- python3 tools/replay_probe.py --workload config4 --profiling 2 > $O/replay_config4_plan.txt 2> $O/trace3.err
python tools/stage_timeline.py $O/trace3/p_kernel_trace.csv > $O/config4_stage_timeline.txt; python tools/stage_span.py $O/trace3/p_kernel_trace.csv > $O/config4_stage_span.txt; grep replay $O/replay_config4_plan.txt | cut -c1-120 >> $O/config4_stage_span.txt; cat $O/config4_stage_span.txt
# 2c. how a resident batch is replayed: plain launches on the forked streams (default), the same as one hipGraph with cross-stream
#     edges (retired in round 6: profiles/r04_replay_modes.txt), one stream as a hipGraph (FSEG_NO_FORK=1), one stream as plain launches -- ms per replay and the
#     host's time inside fseg_run
for e in "-" "FSEG_NO_FORK=1" "FSEG_NO_FORK=1 FSEG_NO_GRAPH=1"; do
  if [ "$e" = "-" ]; then v=""; else v="$e"; fi
  echo "== ${v:-default}"; env $v python tools/replay_probe.py --workload config4 --profiling 0 | grep "replay\|host time" | cut -c1-70
done > $O/replay_modes.txt 2>&1; cat $O/replay_modes.txt
# 3. counters, each in its own pass
for w in config4 config2; do
  for pmc in FETCH_SIZE WRITE_SIZE; do
    FSEG_NO_FORK=1 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmc_${w}_$pmc -o p -- python3 tools/replay_probe.py --workload $w > /dev/null 2> $O/pmc_${w}_$pmc.err
  done
  FSEG_NO_FORK=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${w}_sq -o p -- python3 tools/replay_probe.py --workload $w > /dev/null 2> $O/pmc_${w}_sq.err
  python profiles/pmc_summary.py $O/pmc_${w}_FETCH_SIZE/p_counter_collection.csv $O/pmc_${w}_WRITE_SIZE/p_counter_collection.csv > $O/${w}_pmc_summary.txt
  python tools/sq_summary.py $O/pmc_${w}_sq/p_counter_collection.csv $O/pmc_${w}_sq/p_kernel_trace.csv > $O/${w}_sq_summary.txt
done
python profiles/make_traffic.py $O $P > $O/traffic_summary.txt; cp profiles/traffic.json $O/traffic.json
fi
if [[ "${PART:-all}" =~ ^(all|2)$ ]]; then
# 2d. the stage's timeline on config3 (its chains end with the large class's DPs and the mid class's)
rocprofv3 --kernel-trace --output-format csv -d $O/trace3c -o p -- python3 tools/replay_probe.py --workload config3 --profiling 2 > $O/replay_config3_plan.txt 2> $O/trace3c.err
python tools/stage_timeline.py $O/trace3c/p_kernel_trace.csv > $O/config3_stage_timeline.txt; python tools/stage_span.py $O/trace3c/p_kernel_trace.csv > $O/config3_stage_span.txt; grep replay $O/replay_config3_plan.txt | cut -c1-120 >> $O/config3_stage_span.txt; cat $O/config3_stage_span.txt
rm -rf $O/trace3c
# 4. the other workloads, rows N3 / N4, end to end, the host alone
for w in config2 config3 config5; do python bench.py --workload $w --no-cpu-baseline --no-e2e > $O/${w}_bench.json 2>/dev/null; python profiles/benchsum.py < $O/${w}_bench.json | head -1; done
python bench.py --workload cluster-many > $O/n3_cluster_many_bench.json 2>/dev/null
python bench.py --workload cluster-big --steps 2 --no-cpu-baseline > $O/n3_cluster_big_bench.json 2>/dev/null
python bench.py --workload isoforms > $O/n4_isoforms_bench.json 2>/dev/null
python tools/e2e_bench.py --partitions 4000 --reads 500 --generate-only --keep /dev/shm/e2e_$T > /dev/null
for t in 8 16 32; do python tools/e2e_bench.py --partitions 4000 --reads 500 --threads $t --sidecar off --repeat 2 --timing --keep /dev/shm/e2e_$T >> $O/e2e.log 2>&1; done
rm -rf /dev/shm/e2e_$T
# the host alone (no GPU): the CPU share of this lease, the real main() against a stand-in device on a genome-like split
# directory (24 contig directories), the native parser / writer under P processes x T threads
{ echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)   visible CPUs: $(nproc)"; } > $O/host_ceiling.txt
python tools/host_ceiling.py --partitions 16000 --reads 500 --contigs 24 --workers 1,2,4,8 --label 0,1 --repeat 2 --keep /dev/shm/hc_$T >> $O/host_ceiling.txt 2>/dev/null
python tools/host_scaling_probe.py /dev/shm/hc_$T --procs 1,2,4,8,16,32 --threads 1,4,16 >> $O/host_ceiling.txt 2>&1
python tools/host_scaling_probe.py /dev/shm/hc_$T --procs 8,16 --threads 1,4 --write >> $O/host_ceiling.txt 2>&1
grep -h "nr_throttled\|throttled_usec" /sys/fs/cgroup/cpu.stat >> $O/host_ceiling.txt
rm -rf /dev/shm/hc_$T
grep "e2e\[" $O/e2e.log; cat $O/host_ceiling.txt
fi
if [[ "${PART:-all}" =~ ^(all|1)$ ]]; then
# the driver's command once more, now that profiles/traffic.json carries this library's hash: bench_default.json (again) holds roofline.traffic / valu_util
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; python profiles/benchsum.py < $O/bench_default.json
rm -rf $O/stats/*.csv.tmp
fi
