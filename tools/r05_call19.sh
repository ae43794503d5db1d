#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
python -m pytest tests -m gpu -x -q -k "down or non_finite or golden" 2>&1 | tail -12 | tee $out/r05v_pytest.txt
