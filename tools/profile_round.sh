cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json | python profiles/benchsum.py
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v12_stats -o p -- python3 bench.py > gpurun_out/v12_bench_under_rocprof.json 2> gpurun_out/v12_stats.err
grep "^{" gpurun_out/v12_bench_under_rocprof.json | python profiles/benchsum.py
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/v12_fetch -o p -- python3 bench.py --no-cpu-baseline --steps 10 > /dev/null 2> gpurun_out/v12_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/v12_write -o p -- python3 bench.py --no-cpu-baseline --steps 10 > /dev/null 2> gpurun_out/v12_write.err
python profiles/pmc_summary.py gpurun_out/v12_fetch/p_counter_collection.csv gpurun_out/v12_write/p_counter_collection.csv | head -12
for w in config3 config4 config5; do python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_$w.json | python profiles/benchsum.py | head -1; done
