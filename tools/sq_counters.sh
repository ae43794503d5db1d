#!/bin/bash
# SQ counter passes of the dominant kernel (the level-3 dual launch: conv3d_x3q_kernel<2, *>, tools/x3_dual_one.py): where the waves' cycles go.
#   bash tools/sq_counters.sh r03b      (through gpurun, from the repo root; writes gpurun_out/<tag>_sq_counters.txt)
# Counters (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked: s_waitcnt / barrier) + SQ_WAIT_INST_ANY
# (issue stall) + SQ_ACTIVE_INST_ANY; SQ_WAIT_INST_LDS is a sub-bucket of the issue stalls; LDS array cycles and conflicts.
set -o pipefail
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
  d=$out/${tag}_sq_$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d "$d" -- python3 "$root/tools/x3_dual_one.py" > /dev/null 2>&1 || { echo "pass failed: $set"; continue; }
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3d_x3" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
n = {k: v[1] for k, v in acc.items()}
for k, (s, c) in sorted(acc.items()):
    print(f"{k:28s} {s / max(1, c / max(1, min(n.values()))) / max(1, min(n.values())):16.0f} per launch")
PY
  rm -rf "$d"
done | tee "$out/${tag}_sq_counters.txt"
