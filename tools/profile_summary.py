"""Turn rocprofv3 output directories into the files kept under profiles/.

  python tools/profile_summary.py stats  <kernel_stats.csv> <per-unit divisor> <title>    -> markdown table on stdout
  python tools/profile_summary.py pmc    <tag>=<fetch_dir>,<write_dir> [...] <out.json>   -> PMC traffic json

`pmc`: FETCH_SIZE and WRITE_SIZE come from SEPARATE `rocprofv3 --pmc` passes of the same bench command (the
MI355X guide's HBM recipe); rows are grouped by (kernel, workgroups) and averaged per launch.  Bytes =
(2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts half of a wide coalesced read (calibrated on
adam_kernel, whose algorithmic traffic is known: 4 reads + 3 writes of the flat parameter buffer).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("rac::", "")
    name = re.sub(r"^void ", "", name)
    return name


def stats(path, div, title):
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"## {title}: {total / 1e6 / div:.1f} ms of kernels\n")
    print("| % | calls | ms | avg us | kernel |\n|---|---|---|---|---|")
    for r in rows[:18]:
        t = float(r["TotalDurationNs"])
        print(f"| {100 * t / total:.2f} | {int(r['Calls']) / div:g} | {t / 1e6 / div:.2f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | `{short(r['Name'])[:80]}` |")
    print()


def counter_rows(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            wg = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            key = (re.sub(r"\(.*", "", short(r["Kernel_Name"])), wg)
            agg[key][0] += float(r["Counter_Value"])
            agg[key][1] += 1
    return agg


def pmc(specs, out):
    res = {}
    for spec in specs:
        tag, dirs = spec.split("=")
        fdir, wdir = dirs.split(",")
        fetch, write = counter_rows(fdir, "FETCH_SIZE"), counter_rows(wdir, "WRITE_SIZE")
        rows = []
        for key, (fsum, n) in fetch.items():
            wsum, wn = write.get(key, (0.0, 0))
            if not wn or not key[0].startswith(("igemm", "wgrad", "adam", "slab", "split", "lstm", "transpose")):
                continue
            f_kb, w_kb = fsum / n, wsum / wn
            rows.append({"kernel": key[0], "workgroups": key[1], "launches": n, "FETCH_SIZE_KB_raw": f_kb,
                         "WRITE_SIZE_KB": w_kb, "hbm_side_bytes_per_launch": (2 * f_kb + w_kb) * 1024,
                         "_total": (2 * f_kb + w_kb) * n})
        rows.sort(key=lambda r: -r["_total"])
        for r in rows:
            del r["_total"]
        res[tag] = rows[:12]
    res["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py. Bytes = "
                    "(2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half of a wide coalesced read "
                    "(MI355X_MICROARCH.md HBM section; check on adam_kernel: 4 reads + 3 writes of 954 MB). "
                    "L2<->fabric bytes; Infinity-Cache hits are included.")
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], float(sys.argv[3]), sys.argv[4])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
