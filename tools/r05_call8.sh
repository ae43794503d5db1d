#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
L=rag_amd/lib/librag_amd_diag.so
for ns in 8 4 5 6 7 10 13 16; do RAGMI_X3_NSEG=$ns bash tools/x3q_diag.sh $L "g4 nomain" "0" "0" | sed "s/^/nseg $ns: /"; done 2>&1 | tee $out/r05h_nseg.txt
