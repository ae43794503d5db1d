#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by a few percent, so numbers from different gpurun calls do not compare):
#   bash tools/ab_bench.sh rag_amd/lib/librag_amd_base.so rag_amd/lib/librag_amd.so [rounds]
# alternates `python bench.py --no-cpu-baseline` between the two (RAG_AMD_LIB selects the library) and prints ms/step of each run.
a=$1; b=$2; n=${3:-3}
for i in $(seq 1 "$n"); do
  for lib in "$a" "$b"; do
    RAG_AMD_LIB=$(realpath "$lib") python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])"
  done
done
