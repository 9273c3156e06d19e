"""The lazy zero_grad against the full fill (GPU box): N optimiser steps at cfg2 from the same start on a fixed pool of
synthetic videos, twice with the full fill and once with ops.LAZY_ZERO_GRAD: the lazy run must differ from the full-fill run by no more
than two full-fill runs differ from each other (a weight gradient written over a stale buffer equals one added to zeros;
the step itself is reproducible only up to the order of its fp64 BatchNorm-statistics atomics).   python tools/soak_lazy.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from robot_aware_control_amd import ops, synthetic as syn

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
args = type("A", (), dict(group_norm=False, h48=False, cfg5=False))()


def run(lazy):
    ops.LAZY_ZERO_GRAD = lazy
    torch.manual_seed(0)
    cf, tr = bench.build_train(args, dev)
    pool = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.synth_video(seed=700 + i, T=6, B=16).items()}
            for i in range(3)]
    torch.manual_seed(1)
    hist = [tr._train_step(pool[s % 3])["recon_loss"] for s in range(steps)]
    return hist, tr.model.flat_parameters()[0].clone(), tr.model.flat_parameters()[1].clone()


h0, p0, g0 = run(False)
h0b, p0b, g0b = run(False)   # run-to-run noise of the step itself (fp64 atomics of the BatchNorm statistics)
h1, p1, g1 = run(True)
rel = lambda a, b: float((a - b).norm() / b.norm())
print(f"{steps} steps: recon {h0[0]:.5f} -> {h0[-1]:.5f}", flush=True)
print(f"full fill, two runs : |dp|/|p| {rel(p0b, p0):.2e}  |dg|/|g| {rel(g0b, g0):.2e}  max loss diff {max(abs(a - b) for a, b in zip(h0, h0b)):.2e}")
print(f"lazy vs full fill   : |dp|/|p| {rel(p1, p0):.2e}  |dg|/|g| {rel(g1, g0):.2e}  max loss diff {max(abs(a - b) for a, b in zip(h0, h1)):.2e}")
noise_p, noise_g = max(rel(p0b, p0), 1e-9), max(rel(g0b, g0), 1e-9)
assert rel(p1, p0) <= 5 * noise_p + 1e-7 and rel(g1, g0) <= 5 * noise_g + 1e-6, "lazy zero_grad differs by more than the step's own noise"
