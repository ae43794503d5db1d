#!/bin/bash
# GPU-side durations (rocprofv3 kernel trace) of the level-3 z-marching launches under the DIAG switches of a -DRAGMI_DIAG build:
#   bash tools/x3_diag.sh <lib.so> "0 1 2 4 8 16 ..."     (through gpurun, from the repo root)
# RAGMI_X3_DIAG bits: 1 no stores, 2 no MFMA block, 4 no commit (split + LDS writes), 8 no global loads, 16 operand reads at one address
lib=$(realpath "$1"); root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for d in $2; do
  rm -rf /tmp/x3dg; RAGMI_X3_DIAG=$d RAG_AMD_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/x3dg -- python3 "$root/tools/x3_diag_run.py" > /dev/null 2>&1
  f=$(find /tmp/x3dg -name "*kernel_stats.csv" | head -1)
  echo "X3_DIAG=$d"; python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'conv3d_x3_kernel' in r['Name']: print('  %-50s calls %s avg %.1f us' % (r['Name'][12:62], r['Calls'], float(r['AverageNs'])/1e3))
"
done
