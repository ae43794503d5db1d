#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests -m gpu -x -q -k "x3 or down_sampling or golden" 2>&1 | tail -4 | tee $out/r05c_pytest.txt
grep -q "passed" $out/r05c_pytest.txt && ! grep -q "failed" $out/r05c_pytest.txt || exit 1
bash tools/abn_bench.sh 2 "" rag_amd/lib/librag_amd_noq.so rag_amd/lib/librag_amd_ser.so rag_amd/lib/librag_amd.so rag_amd/lib/librag_amd_prio.so rag_amd/lib/librag_amd_serprio.so 2>&1 | tee $out/r05c_ab.txt
