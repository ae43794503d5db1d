#!/bin/bash
# Same-box A/B of one environment switch:  bash tools/env_ab.sh VAR A B [rounds] [extra bench flags]
var=$1; a=$2; b=$3; n=${4:-3}; extra=${5:-}
for i in $(seq 1 "$n"); do
  for val in "$a" "$b"; do
    env "$var=$val" python bench.py --no-cpu-baseline --no-configs --steps 30 $extra 2>/tmp/ab_err.txt | python -c "
import json,sys
s=sys.stdin.read()
try:
    d=json.loads(s); print('$var=$val', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], 'strict', (d.get('strict_fp32') or {}).get('value_fp32_mfma'), 'e2e', (d.get('end_to_end') or {}).get('value'))
except Exception as e:
    print('$var=$val', 'FAILED', e); print(open('/tmp/ab_err.txt').read()[-1500:])"
  done
done
