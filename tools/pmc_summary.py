"""Summarise rocprofv3 counter passes per kernel (profiling aid).

usage: python tools/pmc_summary.py OUT.json KERNEL_STATS.csv  FETCH_DIR WRITE_DIR
Each *_DIR holds the csv output of one `rocprofv3 --pmc <COUNTER> --output-format csv` pass of the same command
(separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Values are averaged per launch per kernel;
FETCH_SIZE / WRITE_SIZE are in KiB (bench.py applies the gfx950 correction: HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024).
Kernel names are shortened the way bench.py spells them: no `void`, no `ragmi::`, no argument list, no leading `float, `."""
import collections
import csv
import glob
import json
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*\)$", "", name.strip())
    name = name.replace("void ", "").replace("ragmi::", "")
    return name.replace("<float, ", "<").replace("<float>", "")


def counter_avg(directory: str, counter: str):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for d, v in per_dispatch.items():
            acc[names[d]][0] += v
            acc[names[d]][1] += 1
    return {k: s / n for k, (s, n) in acc.items() if n}


def main():
    out, stats, fdir, wdir = sys.argv[1:5]
    res = {}
    for r in csv.DictReader(open(stats)):
        res[short(r["Name"])] = {"avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"])}
    for key, (d, c) in {"fetch_kib": (fdir, "FETCH_SIZE"), "write_kib": (wdir, "WRITE_SIZE")}.items():
        for k, v in counter_avg(d, c).items():
            res.setdefault(k, {})[key] = v
    res = {k: v for k, v in res.items() if "fetch_kib" in v or "write_kib" in v}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("calls", 0)):
        f, w = v.get("fetch_kib", 0.0), v.get("write_kib", 0.0)
        print(f"{k[:70]:70s} avg {v.get('avg_us', 0):9.1f} us  FETCH_SIZE {f:12.1f} KiB  WRITE_SIZE {w:12.1f} KiB  HBM(2F+W) {(2 * f + w) / 1024:9.1f} MiB")


if __name__ == "__main__":
    main()
