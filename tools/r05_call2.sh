#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests -m gpu -x -q -k "x3 or down_sampling or shard or golden" 2>&1 | tail -8 | tee $out/r05b_pytest.txt
grep -q "passed" $out/r05b_pytest.txt && ! grep -q "failed" $out/r05b_pytest.txt || exit 1
RAG_AMD_LIB=$root/rag_amd/lib/librag_amd_diag.so RAGMI_X3_DIAG=32 python tools/x3_stamps.py dual > $out/r05b_x3q_stamps_dual.txt 2>&1 || { tail -20 $out/r05b_x3q_stamps_dual.txt; exit 1; }
cat $out/r05b_x3q_stamps_dual.txt
bash tools/ab_bench.sh rag_amd/lib/librag_amd_noq.so rag_amd/lib/librag_amd.so 2 2>&1 | tee $out/r05b_ab.txt
