# `value` (inputs resident, eight contexts) under library switches, two rounds (tuning only)
for i in 1 2; do for e in "" "FSEG_THR_PART=1" "FSEG_SPLIT_DP=6" "FSEG_SPLIT_DP=3" "FSEG_SPLIT_DP=5" "FSEG_NO_TINY=1" "FSEG_RANGE_SUMS=0"; do
  env $e timeout -k 5 200 python bench.py --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$e]', round(d['value']/1e6,1), 'resident,', round(d['value_h2h']/1e6,1), 'host to host')"
done; done
