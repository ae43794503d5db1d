#!/bin/bash
# Same-box comparison of N builds of the library: bash tools/abn_bench.sh <rounds> "<extra bench.py flags>" lib1.so lib2.so ...
n=$1; extra=$2; shift 2
for i in $(seq 1 "$n"); do
  for lib in "$@"; do
    RAG_AMD_LIB=$(realpath "$lib") python bench.py --no-cpu-baseline --no-configs --steps 30 $extra 2>/tmp/ab_err.txt | python -c "
import json,sys
s=sys.stdin.read()
try:
    d=json.loads(s); print('%-44s' % '$lib', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], 'strict', (d.get('strict_fp32') or {}).get('value_fp32_mfma'), 'e2e', (d.get('end_to_end') or {}).get('value'))
except Exception as e:
    print('$lib', 'FAILED', e); print(open('/tmp/ab_err.txt').read()[-1500:])"
  done
done
