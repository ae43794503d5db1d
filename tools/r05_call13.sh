#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
RAG_AMD_LIB=$root/rag_amd/lib/librag_amd_diag.so RAGMI_X3_DIAG=32 python tools/x3_stamps.py dual > $out/r05m_x3q_stamps_dual.txt 2>&1 || { tail -20 $out/r05m_x3q_stamps_dual.txt; exit 1; }
cat $out/r05m_x3q_stamps_dual.txt
RAG_AMD_LIB=$root/rag_amd/lib/librag_amd_diag.so RAGMI_X3_DIAG=32 python tools/x3_stamps.py stem1 > $out/r05m_x3_stamps_stem1.txt 2>&1
head -14 $out/r05m_x3_stamps_stem1.txt
