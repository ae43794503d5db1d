#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
L=rag_amd/lib/librag_amd_diag.so
{ bash tools/x3q_diag.sh $L "g4 nomain" "0 1 2 4 8 16 3 10 11 27 31" "0"
  bash tools/x3q_diag.sh $L "g4 nomain" "0 2" "256 384"
  bash tools/x3q_diag.sh $L "planes main" "0 1 8"  "0"; } 2>&1 | tee $out/r05f_x3q_diag.txt
