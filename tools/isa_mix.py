"""Instruction mix of a kernel (whole body, or its innermost loop) from hipcc's assembly listing — how the v_mov / s_load /
spill pathologies quoted in DESIGN.md were found.

    python tools/isa_mix.py rag_amd/csrc/conv3d_x3.hip "conv3d_x3_kernel<float, 2, 2, false>" [--loop] [extra hipcc flags ...]

Compiles the file for gfx950 with the Makefile's flags (-S, device only), finds the kernels whose demangled name contains the
pattern and prints, for each: lines, MFMA / VALU / SALU / LDS / VMEM / wait counts and the most frequent opcodes."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main() -> None:
    src, pattern = sys.argv[1], sys.argv[2]
    rest = sys.argv[3:]
    loop = "--loop" in rest
    flags = [f for f in rest if f != "--loop"]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{ROOT}/rag_amd/csrc",
               "-S", "--cuda-device-only", "-o", asm, src] + flags
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        txt = open(asm).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n", txt, re.M):
        name, i = m.group(1), m.end()
        j = txt.find("s_endpgm", i)
        if j < 0:
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if pattern not in dem:
            continue
        body = txt[i:j].split("\n")
        if loop:
            hdr = [n for n, line in enumerate(body) if "Inner Loop Header" in line]
            if hdr:
                body = body[hdr[-1]:]
        ops = collections.Counter()
        for line in body:
            mm = re.match(r"\s+([a-z_0-9]+)", line)
            if mm:
                ops[mm.group(1)] += 1
        grp = collections.Counter()
        for k, v in ops.items():
            if k.startswith("v_mfma"):
                grp["mfma"] += v
            elif k.startswith("v_"):
                grp["valu"] += v
            elif k.startswith(("s_waitcnt", "s_nop", "s_barrier")):
                grp["wait"] += v
            elif k.startswith("s_"):
                grp["salu"] += v
            elif k.startswith("ds_"):
                grp["lds"] += v
            elif k.startswith(("global_", "buffer_", "scratch_", "flat_")):
                grp["vmem"] += v
        print(dem[:110])
        print("   lines", len(body), dict(grp), "spill moves", ops["v_readlane_b32"] + ops["v_writelane_b32"], "scratch",
              sum(v for k, v in ops.items() if k.startswith("scratch_")))
        print("   ", ops.most_common(int(os.environ.get("ISA_TOP", "14"))))


if __name__ == "__main__":
    main()
