for i in 1 2 3; do for e in ${VARIANTS:-"" "FSEG_SPLIT_ALWAYS=0"}; do
  env $e timeout -k 5 200 python bench.py --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$e]', round(d['value']/1e6,1), 'resident,', round(d['value_h2h']/1e6,1), 'host to host')"
done; done
