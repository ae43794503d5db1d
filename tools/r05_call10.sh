#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
L=rag_amd/lib/librag_amd_diag.so
{ bash tools/x3q_diag.sh $L "g4 nomain" "0" "0"
  bash tools/x3q_diag.sh $L "g4 nomain down" "0 64 128 192 1 2 193" "0"; } 2>&1 | tee $out/r05j_x3q_down_diag.txt
