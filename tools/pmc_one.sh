#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the kernels of one script, separate passes (MI355X_MICROARCH.md: FETCH x 2 for wide streaming reads):
#   bash tools/pmc_one.sh tools/bench_disp.py disp_softargmin      (through gpurun, from the repo root)
root=${GRAFT_REPO_ROOT:-$(pwd)}; script=$1; pat=$2
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc1; rocprofv3 --pmc $c --output-format csv -d /tmp/pmc1 -- python3 "$root/$script" > /dev/null 2>&1
  python3 - "$pat" "$c" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/pmc1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]:
            a = acc[r["Kernel_Name"][:70]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in acc.items():
    print(f"{sys.argv[2]:11s} {k:70s} {s / n:12.1f} KiB per launch ({n} launches)")
PY
done
