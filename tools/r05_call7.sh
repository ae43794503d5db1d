#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
for g in 1 0; do
( cd /tmp && export TMPDIR=/tmp && RAGMI_G4=$g rocprofv3 --kernel-trace --stats --output-format csv -d $out/r05g_kt$g -- python3 $root/bench.py --steps 90 --warmup 2 --no-cpu-baseline --no-configs > /dev/null 2>&1 )
python3 tools/step_timeline.py $out/r05g_kt$g > $out/r05g_timeline_g4_$g.txt; rm -rf $out/r05g_kt$g
done
tail -40 $out/r05g_timeline_g4_1.txt
