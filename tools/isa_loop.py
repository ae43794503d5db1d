"""Instruction mix of the smallest loop of a kernel that holds at least N matrix instructions (the plane step of the z-marching
kernels), from hipcc's assembly listing:
    python tools/isa_loop.py rag_amd/csrc/conv3d_x3.hip "conv3d_x3_kernel<float, 2, 2, 1>" [--min-mfma 40] [--dump] [extra hipcc flags]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, pattern = sys.argv[1], sys.argv[2]
    rest = sys.argv[3:]
    dump = "--dump" in rest
    min_mfma = 40
    if "--min-mfma" in rest:
        i = rest.index("--min-mfma"); min_mfma = int(rest[i + 1]); del rest[i:i + 2]
    flags = [f for f in rest if f != "--dump"]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{ROOT}/rag_amd/csrc",
                        "-S", "--cuda-device-only", "-o", asm, src] + flags, check=True, stderr=subprocess.DEVNULL)
        txt = open(asm).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n", txt, re.M):
        name, i = m.group(1), m.end()
        j = txt.find("s_endpgm", i)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if j < 0 or pattern not in dem:
            continue
        body = txt[i:j].split("\n")
        labels = {mm.group(1): n for n, line in enumerate(body) if (mm := re.match(r"(\.LBB\d+_\d+):", line))}
        best = None
        for n, line in enumerate(body):
            mm = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", line) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", line)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
                lo = labels[mm.group(1)]
                nm = sum(1 for q in body[lo:n] if "v_mfma" in q)
                if nm >= min_mfma and (best is None or n - lo < best[1] - best[0]):
                    best = (lo, n)
        if best is None:
            print(dem[:100], ": no loop with", min_mfma, "MFMAs"); continue
        loop = body[best[0]:best[1] + 1]
        ops = collections.Counter()
        for line in loop:
            mm = re.match(r"\s+([a-z_0-9]+)", line)
            if mm:
                ops[mm.group(1)] += 1
        grp = collections.Counter()
        for k, v in ops.items():
            grp["mfma" if k.startswith("v_mfma") else "valu" if k.startswith("v_") else "wait" if k.startswith(("s_waitcnt", "s_nop", "s_barrier"))
                else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other"] += v
        print(dem[:110])
        print("   loop lines", len(loop), dict(grp))
        print("   ", ops.most_common(int(os.environ.get("ISA_TOP", "24"))))
        if dump:
            print("\n".join(loop))


if __name__ == "__main__":
    main()
