"""Soak run of the train step (GPU box): N optimiser steps at cfg2 on a fixed pool of synthetic videos with the fused
optimiser step, then the same with RAC_ADAM_FUSED=0 semantics (ops.ADAM_FUSED off) from the same start: losses finite and
falling, the two parameter sets close (same arithmetic; the operand scales of the weights may differ by a power of two
when a bound crosses one, which changes conv roundings at the 1e-7 level).   python tools/soak_train.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from robot_aware_control_amd import ops, synthetic as syn

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
args = type("A", (), dict(group_norm=False, h48=False, cfg5=False))()


def run(fused):
    ops.ADAM_FUSED = fused
    torch.manual_seed(0)
    cf, tr = bench.build_train(args, dev)
    cf.lr = tr.optimizer.param_groups[0]["lr"] = 3e-4
    pool = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.synth_video(seed=500 + i, T=6, B=16).items()}
            for i in range(4)]
    torch.manual_seed(1)
    hist = []
    for s in range(steps):
        out = tr._train_step(pool[s % 4])
        hist.append(out["recon_loss"])
        if s % 50 == 0 or s == steps - 1:
            print(f"fused={fused} step {s}: recon {out['recon_loss']:.5f} kld {out['kld']:.4f}", flush=True)
    flat = tr.model.flat_parameters()[0].clone()
    assert torch.isfinite(flat).all() and all(h == h for h in hist)
    return hist, flat


h1, p1 = run(True)
h0, p0 = run(False)
assert h1[-1] < h1[0] and h0[-1] < h0[0], (h1[0], h1[-1], h0[0], h0[-1])
rel = float((p1 - p0).norm() / p0.norm())
print(f"{steps} steps: recon {h1[0]:.4f} -> {h1[-1]:.4f} (fused), {h0[0]:.4f} -> {h0[-1]:.4f} (separate passes); "
      f"|p_fused - p_separate| / |p| = {rel:.2e}; final loss difference {abs(h1[-1] - h0[-1]) / h0[-1]:.2e}")
