# `value` (inputs resident) for other context counts / launch modes (tuning only)
for cfg in "8:" "4:" "6:" "12:" "16:" "8:FSEG_NO_GRAPH=1" "8:GPU_MAX_HW_QUEUES=8" "16:GPU_MAX_HW_QUEUES=8" "8:FSEG_SPLIT_ALWAYS=1"; do
  n=${cfg%%:*}; e=${cfg#*:}
  env $e timeout -k 5 200 python bench.py --no-cpu-baseline --no-e2e --no-extras --contexts $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('contexts $n $e:', round(d['value']/1e6,1), 'M reads/s resident,', round(d['value_h2h']/1e6,1), 'host to host')"
done
