#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
for f in 1 0; do
( cd /tmp && export TMPDIR=/tmp && RAGMI_FUSE_STEMS=$f rocprofv3 --kernel-trace --stats --output-format csv -d $out/r05x_kt$f -- python3 $root/bench.py --steps 90 --warmup 2 --no-cpu-baseline --no-configs > /dev/null 2>&1 )
python3 tools/step_timeline.py $out/r05x_kt$f > $out/r05x_timeline_fuse$f.txt; rm -rf $out/r05x_kt$f
head -8 $out/r05x_timeline_fuse$f.txt; tail -1 $out/r05x_timeline_fuse$f.txt
done
