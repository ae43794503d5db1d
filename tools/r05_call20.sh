#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "costvol or g4 or golden or stem" 2>&1 | tail -15 | tee $out/r05w_pytest.txt
grep -q "passed" $out/r05w_pytest.txt && ! grep -q "failed" $out/r05w_pytest.txt || exit 1
for f in 0 1 0 1; do RAGMI_FUSE_STEMS=$f python bench.py --no-cpu-baseline --no-configs --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('FUSE=$f', d['ms_per_step'], d['value'], 'strict', (d.get('strict_fp32') or {}).get('value_fp32_mfma'), 'e2e', (d.get('end_to_end') or {}).get('value'))"; done 2>&1 | tee $out/r05w_ab.txt
