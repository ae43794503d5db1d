#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests -m gpu -x -q -k "deep or golden or x3" 2>&1 | tail -4 | tee $out/r05r_pytest.txt
grep -q "passed" $out/r05r_pytest.txt && ! grep -q "failed" $out/r05r_pytest.txt || exit 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/r05r_kt -- python3 $root/bench.py --steps 90 --warmup 2 --no-cpu-baseline --no-configs > /dev/null 2>&1 )
python3 tools/step_timeline.py $out/r05r_kt > $out/r05r_timeline.txt; rm -rf $out/r05r_kt
cat $out/r05r_timeline.txt
