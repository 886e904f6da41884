# the headline value for other batch sizes / context counts (tuning only; the default is what bench.py reports)
for cfg in "250000 8" "500000 8" "500000 4" "125000 8" "125000 16" "250000 6"; do
  set -- $cfg
  FREDDIE_BENCH_BATCH_READS=$1 timeout -k 5 200 python bench.py --no-cpu-baseline --no-e2e --no-extras --contexts $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 contexts $2:', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],3), 'ms/step')"
done
