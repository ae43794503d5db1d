#!/bin/bash
# GPU-side durations (rocprofv3 kernel trace, 300 back-to-back launches: warm clocks) of the level-3 dual launch on the quad-ring kernel under
# the DIAG switches of a -DRAGMI_DIAG build:   bash tools/x3q_diag.sh <lib.so> "<layout> <nomain|main>" "0 1 2 4 8 16 ..." ["grid sizes"]
lib=$(realpath "$1"); root=${GRAFT_REPO_ROOT:-$(pwd)}; mode=$2
cd /tmp && export TMPDIR=/tmp
for gsz in ${4:-0}; do
for d in $3; do
  rm -rf /tmp/x3dg; RAGMI_X3_GRID=$gsz RAGMI_X3_DIAG=$d RAG_AMD_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/x3dg -- python3 "$root/tools/x3q_diag_run.py" $mode > /dev/null 2>&1
  f=$(find /tmp/x3dg -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'conv3d_x3' in r['Name']: print('mode [$mode] grid $gsz X3_DIAG=$d  %-40s calls %s avg %.1f us min %.1f' % (r['Name'][12:52], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
"
done
done
