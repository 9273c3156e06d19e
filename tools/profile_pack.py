"""Turn the rocprofv3 output of tools/run_profiles.sh into the files kept under profiles/.

    python tools/profile_pack.py gpurun_out/<tag> <tag>

  profiles/<tag>_train_shapes.md / _cem_shapes.md   per-shape roofline tables (tools/shape_profile.py)
  profiles/<tag>_train_kernel_stats.csv / _cem_..    rocprofv3 --stats kernel summaries
  profiles/<tag>_pmc_traffic.json                    L2<->fabric bytes per launch of the dominant kernels
  profiles/<tag>_sq_summary.md                       MFMA-busy %, held clock, LDS bank conflicts, instruction mix

HBM-side bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of a wide coalesced read
(MI355X_MICROARCH.md, HBM section); FETCH / WRITE come from separate --pmc passes.  Effective clock = GRBM_GUI_ACTIVE
/ 8 XCDs / kernel duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)."""
import csv
import glob
import io
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("conv16_tile_kernel", "conv16_rows_kernel", "conv16_rows_persist_kernel", "wgrad16_kernel", "wgrad16_allky_kernel")


def short(name):
    return re.sub(r"\(.*", "", name.replace("rac::", "").replace("void ", ""))


# the HBM-bound tail of both workloads (bench.py's roofline.tail): bytes from the PMC passes, durations from the kernel stats
TAIL = {"train": ("adam_frag_multi_kernel", "adam_ranges_kernel", "bn_bwd_reduce_rows_kernel", "bn_bwd_apply_rows_kernel",
                  "bn_apply_act_rows_kernel", "slab_reduce_stats_rows_kernel", "lstm_cell_bwd_srcs_kernel", "lstm_cell_fwd_kernel",
                  "head_dgrad_kernel"),
        "cem": ("first16_kernel", "head16_kernel", "cem_step_tail_kernel")}


def counters(d, kernels=KERNELS, with_counts=False):
    """{(kernel, workgroups): {counter: mean over that shape's launches}}; `with_counts`: also {"_n": launches of the shape}."""
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith(kernels):
                wg = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
                agg[(k, wg)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
    if with_counts:
        for k, cs in agg.items():
            out[k]["_n"] = max(len(v) for v in cs.values())
    return out


def main(src, tag):
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    for wl, steps, title in (("train", 5, "cfg2 train step (the 5 timed steps of the run)"),
                             ("cem", None, "cfg3 CEM")):
        trace = os.path.join(src, f"stats_{wl}", "run_kernel_trace.csv")
        shapes = os.path.join(src, f"{wl}_shapes.json")
        stats = os.path.join(src, f"stats_{wl}", "run_kernel_stats.csv")
        if os.path.exists(stats):
            shutil.copy(stats, os.path.join(prof, f"{tag}_{wl}_kernel_stats.csv"))
        if os.path.exists(trace) and os.path.exists(shapes):
            window = "adam_frag_multi:5"
            if steps is None:  # CEM: the timed iteration = the last 14 model steps (one cem_step_tail launch each)
                steps, window = 1, "cem_step_tail:14"
                title += " (the timed iteration: 14 model steps; per-iteration numbers)"
            md = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shape_profile.py"), trace, shapes,
                                 str(steps), title, window], capture_output=True, text=True, check=True).stdout
            open(os.path.join(prof, f"{tag}_{wl}_shapes.md"), "w").write(md)
    # ---- fabric traffic
    res = {}
    for name in ("gemm_train", "gemm_cem", "wgrad5", "wgrad3", "train"):
        f, w = counters(os.path.join(src, f"pmc_FETCH_SIZE_{name}")), counters(os.path.join(src, f"pmc_WRITE_SIZE_{name}"))
        rows = []
        for key, c in f.items():
            if key not in w:
                continue
            fk, wk = c.get("FETCH_SIZE", 0.0), w[key].get("WRITE_SIZE", 0.0)
            rows.append({"kernel": key[0], "workgroups": key[1], "FETCH_SIZE_KB_raw": fk, "WRITE_SIZE_KB": wk,
                         "hbm_side_bytes_per_launch": (2 * fk + wk) * 1024})
        res[name] = sorted(rows, key=lambda r: -r["hbm_side_bytes_per_launch"])[:8]
    # ---- the memory-bound tail: bytes per launch (all launches of a kernel name pooled, weighted by their count) and GB/s
    tail = []
    for wl, names in TAIL.items():
        stats = os.path.join(src, f"stats_{wl}", "run_kernel_stats.csv")
        dur = {}
        if os.path.exists(stats):
            for r in csv.DictReader(open(stats)):
                dur[short(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]))
        f = counters(os.path.join(src, f"pmc_FETCH_SIZE_{wl}"), names, with_counts=True)
        w = counters(os.path.join(src, f"pmc_WRITE_SIZE_{wl}"), names)
        for name in names:
            fk = [(k, c) for k, c in f.items() if k[0].startswith(name)]
            if not fk:
                continue
            # a kernel name launched at several grid sizes: the mean over its LAUNCHES -- every shape's mean weighted by its
            # launch count -- which is what the kernel-stats average duration below is the mean over
            n_all = sum(c["_n"] for _, c in fk)
            fetch = sum(c.get("FETCH_SIZE", 0.0) * c["_n"] for _, c in fk) / n_all
            write = sum(w.get(k, {}).get("WRITE_SIZE", 0.0) * c["_n"] for k, c in fk) / n_all
            d = next((v for k, v in dur.items() if k.startswith(name)), None)
            if d is None:
                continue
            nbytes = (2 * fetch + write) * 1024
            tail.append({"workload": wl, "kernel": name, "hbm_side_bytes_per_launch": nbytes, "avg_us": d[0] / 1e3,
                         "launches_in_trace": d[1], "gbps": nbytes / d[0], "frac_of_8TBps": nbytes / d[0] / 8000.0})
    res["tail"] = tail
    res["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes. Bytes = (2*FETCH_SIZE + "
                    "WRITE_SIZE)*1024 (gfx950 FETCH_SIZE reports half of a wide coalesced read). L2<->fabric bytes; "
                    "Infinity-Cache hits are included. gemm_train / gemm_cem: tools/bench_gemm.py fwd at M=1024 / 64000; "
                    "wgrad5 / wgrad3: the time-batched (T=5, B=16) ConvLSTM weight gradients.  tail: the memory-bound kernels of both workloads -- "
                    "bytes per launch averaged over a kernel's launch shapes (PMC passes of bench.py), average duration from the "
                    "kernel-trace run of the same build (weight gradients in order), GB/s = bytes / duration.")
    json.dump(res, open(os.path.join(prof, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    # ---- SQ summary
    out = io.StringIO()
    out.write(f"# {tag}: SQ / GRBM counters of the conv kernels (tools/run_profiles.sh, separate --pmc passes)\n\n")
    out.write("| run | kernel (workgroups) | MFMA busy | wave cycles: active / wait-inst / wait | LDS bank conflicts / LDS active "
              "| VALU per MFMA | LDS instr per MFMA | VMEM reads per MFMA |\n|---|---|---|---|---|---|---|---|\n")
    names = {1: "gate GEMM fwd M=1024 (bench_gemm)", 2: "gate GEMM fwd M=64000 (bench_gemm)", 3: "wgrad k=5, T=5 B=16",
             4: "wgrad k=3, T=5 B=16", 5: "CEM iteration (bench.py --workload cem)"}
    for i in range(1, 6):
        a, b = counters(os.path.join(src, f"sq1_{i}")), counters(os.path.join(src, f"sq2_{i}"))
        for key in sorted(a, key=lambda k: -a[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:3]:
            c, d = a[key], b.get(key, {})
            gui = c.get("GRBM_GUI_ACTIVE", 0) / 8
            busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * gui) if gui else 0
            wc = c.get("SQ_WAVE_CYCLES", 1)
            mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 16  # 16 cycles per v_mfma_f32_16x16x32_f16
            lds_act = d.get("SQ_LDS_IDX_ACTIVE", 0)
            out.write(f"| {names[i]} | `{key[0]}` ({key[1]}) | {100 * busy:.1f} % | "
                      f"{100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0f} / {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} / "
                      f"{100 * c.get('SQ_WAIT_ANY', 0) / wc:.0f} % | {c.get('SQ_LDS_BANK_CONFLICT', 0):.3g} / {lds_act:.3g} | "
                      f"{(d.get('SQ_INSTS_VALU', 0) - mfma) / mfma if mfma else 0:.2f} | "
                      f"{d.get('SQ_INSTS_LDS', 0) / mfma if mfma else 0:.3f} | {d.get('SQ_INSTS_VMEM_RD', 0) / mfma if mfma else 0:.3f} |\n")
    out.write("\nMFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the share of matrix-pipe "
              "cycles that carry an MFMA; the rest of the gap to the 2.5 PFLOP/s peak is the clock the chip holds under this "
              "load (achieved TFLOP/s = 833.3 x busy x clock / 2.4 GHz; clocks in the shapes tables' durations).\n")
    open(os.path.join(prof, f"{tag}_sq_summary.md"), "w").write(out.getvalue())
    print("packed", tag)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
