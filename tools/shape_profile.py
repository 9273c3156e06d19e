"""Per-shape roofline table from a rocprofv3 kernel trace + the launch log of the same run.

  RAC_SHAPE_LOG=shapes.json rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 bench.py ...
  python tools/shape_profile.py DIR/run_kernel_trace.csv shapes.json STEPS "title"  > profiles/<tag>_shapes.md

The i-th dispatch of a kernel family in the trace is the i-th launch the host logged for that family (one stream, in
order).  Rows: one per (kernel, mode, k, M, N, K); TF = algorithmic FLOP (2 M N K, padding taps included) / duration;
frac = TF / the pipe's peak (fp16 MFMA at 3 products per fp32 product: 2500/3 = 833.3; exact-fp32 MFMA: 157.3).
Kernels outside the three conv families are summed by name below the table."""
import csv
import json
import re
import sys
from collections import OrderedDict, defaultdict

FAMILIES = [("conv16", re.compile(r"conv16_(tile|rows|rows_persist)_kernel")), ("wgrad16", re.compile(r"wgrad16_(allky_)?kernel")),
            ("igemm", re.compile(r"igemm_\w*kernel"))]


def main(trace, shapes, steps, title, window=None):
    """`window` = "<kernel substring>:<count>": only the launches after the (count + 1)-th last launch of that kernel up to
    its last one are tabulated (train: "adam_frag_multi:5" = the five timed steps; planner: "cem_step_tail:14" = the timed
    iteration) -- model construction, allocator priming and warm-up launches stay out of the per-step numbers."""
    log = defaultdict(list)
    for r in json.load(open(shapes)):
        log[r["family"]].append(r)
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Dispatch_Id"]))
    lo, hi = 0, len(rows)
    if window:
        sub, cnt = window.rsplit(":", 1)
        marks = [i for i, r in enumerate(rows) if sub in r["Kernel_Name"]]
        lo, hi = marks[-1 - int(cnt)] + 1, marks[-1] + 1
    seen = defaultdict(int)
    agg = OrderedDict()
    other = defaultdict(lambda: [0, 0.0])
    total = 0.0
    n_rows = 0
    for i, r in enumerate(rows):
        name = r["Kernel_Name"]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3  # us
        fam = next((f for f, rx in FAMILIES if rx.search(name)), None)
        inside = lo <= i < hi
        if not inside:  # (the launch log stays aligned: every conv launch of the run consumes its record)
            if fam is not None and seen[fam] < len(log[fam]):
                seen[fam] += 1
            continue
        total += dur
        n_rows += 1
        if fam is None or seen[fam] >= len(log[fam]):
            short = re.sub(r"\(.*", "", name.replace("rac::", "").replace("void ", ""))[:60]
            other[short][0] += 1
            other[short][1] += dur
            continue
        rec = log[fam][seen[fam]]
        seen[fam] += 1
        kern = re.sub(r"\(.*", "", name.replace("rac::", "").replace("void ", ""))
        key = (kern, rec["mode"], rec["k"], rec["M"], rec["N"], rec["K"])
        a = agg.setdefault(key, {"n": 0, "us": 0.0, "flop": rec["flop"], "peak": rec["peak"]})
        a["n"] += 1
        a["us"] += dur
    for fam, _ in FAMILIES:
        if seen[fam] != len(log[fam]):
            print(f"<!-- warning: {fam}: {seen[fam]} dispatches in the trace, {len(log[fam])} launches logged -->")
    print(f"## {title}: {total / 1e3 / steps:.2f} ms of kernels per step ({steps} steps profiled)\n")
    print("| kernel | mode | k | M | N | K | launches/step | avg us | ms/step | TFLOP/s | of pipe peak |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    conv_ms = conv_flop = 0.0
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
        kern, mode, k, M, N, K = key
        avg = a["us"] / a["n"]
        tf = a["flop"] / (avg * 1e-6) / 1e12
        conv_ms += a["us"] / 1e3 / steps
        conv_flop += a["flop"] * a["n"] / steps
        print(f"| `{kern}` | {mode} | {k} | {M} | {N} | {K} | {a['n'] / steps:g} | {avg:.1f} | {a['us'] / 1e3 / steps:.3f} | "
              f"{tf:.0f} | {tf / a['peak']:.2f} ({a['peak']:.0f}) |")
    print(f"\nconv kernels: {conv_ms:.2f} ms/step for {conv_flop / 1e12:.2f} algorithmic TFLOP = "
          f"{conv_flop / 1e12 / (conv_ms * 1e-3):.0f} TFLOP/s\n")
    print("| other kernel | launches/step | ms/step |\n|---|---|---|")
    for name, (n, us) in sorted(other.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"| `{name}` | {n / steps:g} | {us / 1e3 / steps:.3f} |")
    rest = sorted(other.items(), key=lambda kv: -kv[1][1])[25:]
    if rest:
        print(f"| ({len(rest)} more) | {sum(v[0] for _, v in rest) / steps:g} | {sum(v[1] for _, v in rest) / 1e3 / steps:.3f} |")
    print(f"\nlaunches per step: {n_rows / steps:.0f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
