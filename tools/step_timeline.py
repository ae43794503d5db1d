"""Kernel-by-kernel timeline of ONE headline forward pass from a rocprofv3 kernel trace.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline
    python tools/step_timeline.py <dir> > profiles/rNN_step_timeline.txt

The pass is found as the last run of dispatches that starts with costvol_stem_planes_mfma_kernel (the headline's first kernel under
the default precision; or the kernel named by a second argument, e.g. costvol_stem_planes_kernel for the strict-fp32 pass) and
ends with the next disp_softargmin kernel.
"""
import csv
import glob
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("ragmi::", "")
    return re.sub(r"\(.*$", "", name)


def main(root: str, first: str = "costvol_stem_planes_mfma_kernel", must: str = "") -> None:
    files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + root)
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if first in r[2]]
    ends = [i for i, r in enumerate(rows) if "disp_softargmin" in r[2]]
    if not starts or not ends:
        raise SystemExit("no forward pass found")
    # the last complete pass of that kind (first kernel ... the next soft-argmin) [that holds a kernel whose name contains `must`]
    s = e = None
    for cand in reversed(starts):
        later = [i for i in ends if i > cand]
        if later and (not must or any(must in r[2] for r in rows[cand:later[0] + 1])):
            s, e = cand, later[0]
            break
    if s is None:
        raise SystemExit("no complete forward pass found")
    t0 = rows[s][0]
    print("start_us  dur_us  gap_us  kernel")
    prev_end = t0
    total = 0.0
    for st, en, name in rows[s:e + 1]:
        print(f"{(st - t0) / 1e3:8.1f} {(en - st) / 1e3:7.1f} {(st - prev_end) / 1e3:7.1f}  {short(name)}")
        prev_end = en
        total += (en - st) / 1e3
    print(f"# wall {(rows[e][1] - t0) / 1e3:.1f} us, kernel time {total:.1f} us, {e - s + 1} launches")


if __name__ == "__main__":
    # optional second argument: name fragment of the pass's FIRST kernel (conv2d_k3_strided_kernel = the end-to-end pass from images)
    # optional third argument: a name fragment some kernel of the pass must contain (e.g. bf16 for the bf16-storage pass)
    main(sys.argv[1] if len(sys.argv) > 1 else ".", *(sys.argv[2:4]))
