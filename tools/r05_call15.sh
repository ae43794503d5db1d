#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $out/r05q_pytest.txt
grep -q "passed" $out/r05q_pytest.txt && ! grep -q "failed" $out/r05q_pytest.txt || exit 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/r05q_kt -- python3 $root/bench.py --steps 90 --warmup 2 --no-cpu-baseline --no-configs > /dev/null 2>&1 )
python3 tools/step_timeline.py $out/r05q_kt > $out/r05q_timeline.txt; rm -rf $out/r05q_kt
cat $out/r05q_timeline.txt
python bench.py --no-cpu-baseline --no-configs --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], 'strict', (d.get('strict_fp32') or {}).get('value_fp32_mfma'), 'e2e', (d.get('end_to_end') or {}).get('value'))"
