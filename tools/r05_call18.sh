#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests -m gpu -x -q -k "g4 or x3 or down_sampling or golden" 2>&1 | tail -4 | tee $out/r05u_pytest.txt
grep -q "passed" $out/r05u_pytest.txt && ! grep -q "failed" $out/r05u_pytest.txt || exit 1
bash tools/abn_bench.sh 3 "" rag_amd/lib/librag_amd_base.so rag_amd/lib/librag_amd.so 2>&1 | tee $out/r05u_ab.txt
