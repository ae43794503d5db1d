#!/bin/bash
# round 5, GPU call 1: stamps + clock of the level-3 launches, A/B of the note_overflow fast path, GUI_ACTIVE clock, x3 tests
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
RAG_AMD_LIB=$root/rag_amd/lib/librag_amd_diag.so RAGMI_X3_DIAG=32 python tools/x3_stamps.py dual > $out/r05a_x3_stamps_dual.txt 2>&1 || { tail -20 $out/r05a_x3_stamps_dual.txt; exit 1; }
cat $out/r05a_x3_stamps_dual.txt
RAG_AMD_LIB=$root/rag_amd/lib/librag_amd_diag.so RAGMI_X3_DIAG=32 python tools/x3_stamps.py stem1 > $out/r05a_x3_stamps_stem1.txt 2>&1 || exit 1
cat $out/r05a_x3_stamps_stem1.txt
bash tools/ab_bench.sh rag_amd/lib/librag_amd_base.so rag_amd/lib/librag_amd.so 2 2>&1 | tee $out/r05a_ab.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/r05a_gui -- python3 $root/tools/x3_dual_one.py > /dev/null 2>&1 )
python3 - <<'PY'
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
d = root + "/gpurun_out/r05a_gui"
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3d_x3_kernel" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3d_x3_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            us = dur.get(r["Dispatch_Id"])
            if us: print(f"dispatch {r['Dispatch_Id']}: {us:.1f} us, GRBM_GUI_ACTIVE {float(r['Counter_Value']):.0f} -> {float(r['Counter_Value']) / 8 / us:.0f} MHz")
PY
python -m pytest tests -m gpu -x -q -k "x3 or down_sampling or shard" 2>&1 | tail -5
