#!/bin/bash
# Kernel-by-kernel timeline of one forward pass:  bash tools/timeline_one.sh <tag> ["extra bench.py flags"] [first kernel of a pass] [fragment a kernel of the pass must contain]
tag=${1:-tl}; extra=${2:-}; first=${3:-costvol_stem_planes}; must=${4:-}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_kt" -- python3 "$root/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs $extra > /dev/null 2>&1 || exit 1
python3 "$root/tools/step_timeline.py" "$out/${tag}_kt" "$first" $must > "$out/${tag}_step_timeline.txt"
rm -rf "$out/${tag}_kt"
cat "$out/${tag}_step_timeline.txt"
