#!/bin/bash
# GPU-side durations (rocprofv3 kernel trace) of the deep-level dual launches under the DIAG switches of a -DRAGMI_DIAG build:
#   bash tools/xd_diag.sh <lib.so> "0 2 4 6 15"     (through gpurun, from the repo root)
# RAGMI_XD_DIAG bits: 1 no stores, 2 no MFMA block, 4 no staging, 8 no weight-fragment copy.  Eager timing from Python is
# host-bound below ~15 us per call, so only the trace's durations mean anything for these kernels.
lib=$(realpath "$1"); root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for d in $2; do
  rm -rf /tmp/xdd; RAGMI_XD_DIAG=$d RAG_AMD_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xdd -- python3 "$root/tools/bench_deep.py" > /dev/null 2>&1
  f=$(find /tmp/xdd -name "*kernel_stats.csv" | head -1)
  echo "XD_DIAG=$d"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'x3d_kernel' in r['Name']: print('  %-60s calls %s avg %.1f us' % (r['Name'][13:73], r['Calls'], float(r['AverageNs'])/1e3))
"
done
