"""Who launches what in one cfg2 train step: rac_* C-ABI calls and ATen ops by call site.
    python tools/launch_census.py [out.md]          (on the GPU box)"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench
from robot_aware_control_amd import _lib, ops, synthetic as syn
from robot_aware_control_amd.trainer import PredictionTrainer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("aten.empty", "aten.view", "aten.as_strided", "aten.permute", "aten.reshape", "aten.detach", "aten.alias",
        "aten.slice", "aten.select", "aten.t.", "aten.transpose", "aten.expand", "aten._unsafe_view", "aten.unsqueeze",
        "aten.squeeze", "aten.unbind", "aten.split", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.narrow",
        "aten.lift_fresh", "aten.set_", "aten.record_stream", "aten._local_scalar", "aten.new_empty", "aten.chunk")


def site(depth=3):
    out = []
    for fr in reversed(traceback.extract_stack()[:-2]):
        if fr.filename.startswith(ROOT) and "launch_census" not in fr.filename and not fr.filename.endswith("_lib.py"):
            out.append(f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}")
            if len(out) == depth:
                break
    return " <- ".join(out) if out else "autograd engine / other"


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.aten = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            self.aten[(name, site())] += 1
        return func(*args, **(kwargs or {}))


def main():
    dev = torch.device("cuda:0")
    cf = bench.namespace(dev, lstm_group_norm=False)
    tr = PredictionTrainer(cf)
    tr.model.train()
    B, T = cf.batch_size, cf.n_past + cf.n_future
    data = {k: (v.to(dev) if torch.is_tensor(v) else v)
            for k, v in syn.synth_video(seed=100, T=T, B=B, H=cf.image_height, W=cf.image_width).items()}
    for _ in range(3):
        tr._train_step(data)
    torch.cuda.synchronize()
    rac = collections.Counter()
    real_call = _lib.call

    def counting_call(name, *a):
        rac[(name, site())] += 1
        return real_call(name, *a)
    ops.call = counting_call
    _lib.call = counting_call
    census = Census()
    with census:
        tr._train_step(data)
    torch.cuda.synchronize()
    if os.environ.get("CENSUS_MEMCPY"):  # who issues the device-side copies (hipMemcpyAsync blit kernels)?
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            tr._train_step(data)
            torch.cuda.synchronize()
        seen = collections.Counter()
        for ev in prof.events():
            for k in getattr(ev, "kernels", []) or []:
                if "copy" in k.name.lower() or "memcpy" in k.name.lower() or "fill" in k.name.lower():
                    st = [f for f in (ev.stack or []) if "robot_aware_control_amd" in f or "bench" in f][:3]
                    seen[(k.name[:40], ev.name, " <- ".join(st))] += 1
        for (kn, en, st), c in seen.most_common(40):
            print(f"{c:4d}  {kn:40s}  {en:30s}  {st}")
    lines = ["# launch census, one cfg2 train step", "", "## rac_* calls", "", "| calls | entry | site |", "|---|---|---|"]
    by_name = collections.Counter()
    for (n, s), c in rac.items():
        by_name[n] += c
    for (n, s), c in sorted(rac.items(), key=lambda kv: -kv[1]):
        lines.append(f"| {c} | {n} | {s} |")
    lines += ["", f"total rac calls: {sum(rac.values())}", "", "| calls | entry |", "|---|---|"]
    lines += [f"| {c} | {n} |" for n, c in by_name.most_common()]
    lines += ["", "## ATen ops (views / allocations excluded)", "", "| calls | op | site |", "|---|---|---|"]
    for (n, s), c in sorted(census.aten.items(), key=lambda kv: -kv[1]):
        lines.append(f"| {c} | {n} | {s} |")
    lines.append(f"\ntotal ATen ops listed: {sum(census.aten.values())}")
    out = "\n".join(lines)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
