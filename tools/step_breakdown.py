"""Per-kernel time of the TIMED train steps of a rocprofv3 kernel trace (tools/kstats.sh): the launches between the last
N + 1 optimiser launches, so that model construction and allocator priming do not pollute the per-step numbers.
    python tools/step_breakdown.py gpurun_out/<tag>/run_kernel_trace.csv [steps=5]"""
import collections
import csv
import re
import sys


def main(path, nsteps=5):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adam_frag_multi" in r["Kernel_Name"]]
    seg = rows[idx[-1 - nsteps] + 1:idx[-1] + 1]
    short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("rac::", ""))[:60]
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in seg:
        k = short(r["Kernel_Name"])
        tot[k][0] += 1
        tot[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6 / nsteps
    ksum = sum(v[1] for v in tot.values()) / 1e3 / nsteps
    print(f"wall/step {span:.2f} ms, kernel sum/step {ksum:.2f} ms, launches/step {len(seg) / nsteps:.0f}")
    conv = other = 0.0
    for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        if any(s in k for s in ("conv16", "wgrad16", "igemm")):
            conv += t
        else:
            other += t
        print(f"{t / 1e3 / nsteps:8.3f} ms {n / nsteps:6.1f}  {t / n:8.1f} us  {k}")
    print(f"conv kernels {conv / 1e3 / nsteps:.2f} ms/step, everything else {other / 1e3 / nsteps:.2f} ms/step")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5)
