#!/bin/bash
# Per-kernel averages (rocprofv3 --stats) of a bench.py run:  bash tools/stats_one.sh <tag> ["extra bench.py flags"] [rows]
tag=${1:-st}; extra=${2:-}; n=${3:-40}
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_kt" -- python3 "$root/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs $extra > /dev/null 2>&1 || exit 1
find "$out/${tag}_kt" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_kernel_stats.csv" \;
rm -rf "$out/${tag}_kt"
python3 - "$out/${tag}_kernel_stats.csv" "$n" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
    name = re.sub(r"^void ", "", r["Name"]).replace("ragmi::", "")
    name = re.sub(r"\(.*$", "", name)
    print(name[:90].ljust(90), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), ("%.2f" % float(r["Percentage"])).rjust(7))
PY
