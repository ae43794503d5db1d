#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests -m gpu -x -q -k "g4 or x3 or down_sampling or golden" 2>&1 | tail -4 | tee $out/r05i_pytest.txt
grep -q "passed" $out/r05i_pytest.txt && ! grep -q "failed" $out/r05i_pytest.txt || exit 1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/r05i_kt -- python3 $root/bench.py --steps 90 --warmup 2 --no-cpu-baseline --no-configs > /dev/null 2>&1 )
python3 tools/step_timeline.py $out/r05i_kt > $out/r05i_timeline.txt; rm -rf $out/r05i_kt
head -8 $out/r05i_timeline.txt; tail -1 $out/r05i_timeline.txt
