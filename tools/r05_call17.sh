#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
bash tools/abn_bench.sh 2 "" rag_amd/lib/librag_amd.so rag_amd/lib/librag_amd_ph30.so rag_amd/lib/librag_amd_ph60.so rag_amd/lib/librag_amd_ph100.so 2>&1 | tee $out/r05t_ab.txt
