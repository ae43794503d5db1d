#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests -m gpu -x -q -k "g4 or x3 or down_sampling or golden or costvol" 2>&1 | tail -12 | tee $out/r05e_pytest.txt
grep -q "passed" $out/r05e_pytest.txt && ! grep -q "failed" $out/r05e_pytest.txt || exit 1
for g in 0 1 0 1; do RAGMI_G4=$g python bench.py --no-cpu-baseline --no-configs --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('G4=$g', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], 'strict', (d.get('strict_fp32') or {}).get('value_fp32_mfma'), 'e2e', (d.get('end_to_end') or {}).get('value'), 'epe', d.get('epe_gpu_vs_cpu_px'))"; done 2>&1 | tee $out/r05e_ab.txt
