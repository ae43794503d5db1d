#!/bin/bash
# Collect the per-round evidence set on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02n
# Writes gpurun_out/<tag>_*: the default bench line, the rocprofv3 kernel-trace stats of the same command, separate --pmc
# FETCH_SIZE / WRITE_SIZE passes summarised per kernel, the kernel-by-kernel timeline of one forward pass, and the other
# configurations' bench lines.  Copy what is to be judged into profiles/.
set -o pipefail
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd "$root" || exit 1
python bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.stderr.txt" || exit 1
echo "[collect] bench done"; cat "$out/${tag}_bench.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_kt" -- python3 "$root/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs > /dev/null 2>&1 || exit 1
find "$out/${tag}_kt" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_bench_kernel_stats.csv" \;
python3 "$root/tools/step_timeline.py" "$out/${tag}_kt" > "$out/${tag}_step_timeline.txt"
rm -rf "$out/${tag}_kt"
echo "[collect] kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_f" -- python3 "$root/bench.py" --graph 0 --steps 3 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2>&1 || exit 1
echo "[collect] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_pmc_w" -- python3 "$root/bench.py" --graph 0 --steps 3 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2>&1 || exit 1
echo "[collect] WRITE_SIZE pass done"
python3 "$root/tools/pmc_summary.py" "$out/${tag}_pmc_summary.json" "$out/${tag}_bench_kernel_stats.csv" "$out/${tag}_pmc_f" "$out/${tag}_pmc_w" > "$out/${tag}_pmc_summary.txt"
rm -rf "$out/${tag}_pmc_f" "$out/${tag}_pmc_w"
# the strict-fp32 path's own counter passes (bench.py's strict_fp32.roofline.traffic reads this summary)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_skt" -- python3 "$root/bench.py" --precision fp32 --steps 10 --warmup 2 --no-cpu-baseline --no-configs > /dev/null 2>&1 || exit 1
find "$out/${tag}_skt" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_strict_kernel_stats.csv" \;
python3 "$root/tools/step_timeline.py" "$out/${tag}_skt" costvol_stem_planes_kernel > "$out/${tag}_strict_step_timeline.txt"
rm -rf "$out/${tag}_skt"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_spmc_f" -- python3 "$root/bench.py" --precision fp32 --graph 0 --steps 3 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_spmc_w" -- python3 "$root/bench.py" --precision fp32 --graph 0 --steps 3 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2>&1 || exit 1
python3 "$root/tools/pmc_summary.py" "$out/${tag}_strict_pmc_summary.json" "$out/${tag}_strict_kernel_stats.csv" "$out/${tag}_spmc_f" "$out/${tag}_spmc_w" > "$out/${tag}_strict_pmc_summary.txt"
rm -rf "$out/${tag}_spmc_f" "$out/${tag}_spmc_w"
echo "[collect] strict fp32 passes done"
cd "$root" || exit 1
python bench.py --dtype bf16 --batch 8 --no-cpu-baseline > "$out/${tag}_bench_config3_bf16_b8.json" 2> "$out/${tag}_bench_config3_bf16_b8.stderr.txt" || exit 1
python bench.py --batch 8 --no-cpu-baseline > "$out/${tag}_bench_f32_b8.json" 2>/dev/null || exit 1
python bench.py --hw 480x960 --batch 8 --no-cpu-baseline > "$out/${tag}_bench_config4_480x960_b8.json" 2>/dev/null || exit 1
python bench.py --train --no-cpu-baseline > "$out/${tag}_train_bench.json" 2> "$out/${tag}_train_bench.stderr.txt" || exit 1
echo "[collect] variants done"
for f in bench_config3_bf16_b8 bench_f32_b8 bench_config4_480x960_b8 train_bench; do python3 -c "import json,sys; d=json.load(open('$out/${tag}_$f.json')); print('$f', d['value'], d['unit'], d['ms_per_step'])"; done
