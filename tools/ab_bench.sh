#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by a few percent, so numbers from different gpurun calls do not compare):
#   bash tools/ab_bench.sh rag_amd/lib/librag_amd_base.so rag_amd/lib/librag_amd.so [rounds] [extra bench.py flags]
# alternates `python bench.py --no-cpu-baseline --no-configs` between the two (RAG_AMD_LIB selects the library) and prints ms/step of each run.
a=$1; b=$2; n=${3:-3}; extra=${4:-}
for i in $(seq 1 "$n"); do
  for lib in "$a" "$b"; do
    RAG_AMD_LIB=$(realpath "$lib") python bench.py --no-cpu-baseline --no-configs --steps 30 $extra 2>/tmp/ab_err.txt | python -c "
import json,sys
s=sys.stdin.read()
try:
    d=json.loads(s); print('$lib', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], 'strict', (d.get('strict_fp32') or {}).get('value_fp32_mfma'), 'e2e', (d.get('end_to_end') or {}).get('value'))
except Exception as e:
    print('$lib', 'FAILED', e); print(open('/tmp/ab_err.txt').read()[-1500:])"
  done
done
