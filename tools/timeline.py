"""Per-stream timeline of ONE train step from a rocprofv3 kernel trace (streams on: no RAC_WGRAD_STREAM=0):

    rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 bench.py --workload train ...
    python tools/timeline.py DIR/.../run_kernel_trace.csv [window kernel substring, default adam_frag_multi] > profiles/<tag>_timeline.md

Takes the last complete step (between the last two launches of the window kernel on the main queue), and prints per queue:
kernel count, busy time, first start and last end relative to the step's start; the main queue's idle gaps; for every queue
but the busiest (the side streams) its spans -- when they start and end against the main queue's phases -- and how long the
main queue ran past the last data-gradient conv (what the side stream's tail exposes)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    return re.sub(r"\(.*", "", name.replace("rac::", "").replace("void ", ""))[:48]


def main(path, mark="adam_frag_multi"):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
    # (one optimiser launch per step on the main queue; with the late update overlapped a second one on the side queue)
    by_q = defaultdict(list)
    for i in marks:
        by_q[rows[i]["Queue_Id"]].append(i)
    mainq = max(by_q, key=lambda q: len(by_q[q]))
    mm = by_q[mainq]
    lo, hi = mm[-2] + 1, mm[-1] + 1
    t0 = int(rows[lo]["Start_Timestamp"])
    step = rows[lo:hi]
    t_end = max(int(r["End_Timestamp"]) for r in step)
    print(f"## one train step on the GPU's queues: {(t_end - t0) / 1e6:.2f} ms from the first kernel's start to the last kernel's end\n")
    qs = defaultdict(list)
    for r in step:
        qs[r["Queue_Id"]].append(r)
    print("| queue | kernels | busy ms | first start ms | last end ms | role |")
    print("|---|---|---|---|---|---|")
    order = sorted(qs, key=lambda q: -len(qs[q]))
    for q in order:
        rs = qs[q]
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
        role = "main stream" if q == order[0] else "side stream"
        print(f"| {q} | {len(rs)} | {busy:.2f} | {(int(rs[0]['Start_Timestamp']) - t0) / 1e6:.2f} | "
              f"{(max(int(r['End_Timestamp']) for r in rs) - t0) / 1e6:.2f} | {role} |")
    main_rows = qs[order[0]]
    gaps = []
    for a, b in zip(main_rows[:-1], main_rows[1:]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if g > 0:
            gaps.append((g, short(a["Kernel_Name"]), short(b["Kernel_Name"]), (int(a["End_Timestamp"]) - t0) / 1e6))
    tot_gap = sum(g for g, *_ in gaps) / 1e6
    print(f"\nmain stream: {len(main_rows)} kernels, idle between them {tot_gap:.2f} ms in {len(gaps)} gaps "
          f"(median {sorted(g for g, *_ in gaps)[len(gaps) // 2] / 1e3:.1f} us); the ten longest:\n")
    print("| gap us | at ms | after | before |\n|---|---|---|---|")
    for g, a, b, at in sorted(gaps, reverse=True)[:10]:
        print(f"| {g / 1e3:.1f} | {at:.2f} | `{a}` | `{b}` |")
    for q in order[1:]:
        rs = qs[q]
        print(f"\nqueue {q} (side): spans of back-to-back kernels (gap < 50 us)\n")
        print("| start ms | end ms | kernels | busy ms | first | last |\n|---|---|---|---|---|---|")
        span = [rs[0]]
        def flush(span):
            busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in span) / 1e6
            print(f"| {(int(span[0]['Start_Timestamp']) - t0) / 1e6:.2f} | {(int(span[-1]['End_Timestamp']) - t0) / 1e6:.2f} | {len(span)} | "
                  f"{busy:.2f} | `{short(span[0]['Kernel_Name'])}` | `{short(span[-1]['Kernel_Name'])}` |")
        for r in rs[1:]:
            if int(r["Start_Timestamp"]) - int(span[-1]["End_Timestamp"]) > 50000:
                flush(span)
                span = []
            span.append(r)
        flush(span)
    # the tail: main-queue time behind its last conv (data gradient) until the step's last kernel anywhere
    convs = [r for r in main_rows if "conv16" in r["Kernel_Name"] or "igemm" in r["Kernel_Name"]]
    if convs:
        last_conv = int(convs[-1]["End_Timestamp"])
        side_end = max((int(r["End_Timestamp"]) for q in order[1:] for r in qs[q]), default=last_conv)
        print(f"\nlast data-gradient conv on the main stream ends at {(last_conv - t0) / 1e6:.2f} ms; the side streams' last kernel at "
              f"{(side_end - t0) / 1e6:.2f} ms; the step's last kernel at {(t_end - t0) / 1e6:.2f} ms")


if __name__ == "__main__":
    main(*sys.argv[1:])
