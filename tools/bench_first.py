"""Micro-benchmark of the frozen model's first encoder layer, both forms (GPU box): python tools/bench_first.py [B] [Cm]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robot_aware_control_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
Cm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
img = torch.rand(B, 3, 64, 64, device=dev)
mask = (torch.rand(B, Cm, 64, 64, device=dev) > 0.7).float() if Cm else None
w = (torch.randn(64, 3 + Cm, 3, 3, device=dev) * 0.2).contiguous(memory_format=torch.channels_last)
scale, shift = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.3
for name, mfma in (("matrix pipe (rac_first_layer_fwd_split)", True), ("exact-fp32 FMAs (rac_first_layer_fwd)", False)):
    ops.FIRST_MFMA = mfma
    for _ in range(3):
        ops.first_layer_frozen(img, None, mask, w, scale, shift)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = ops.first_layer_frozen(img, None, mask, w, scale, shift)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}, {3 + Cm} planes: {ms * 1e3:.0f} us  ({out.numel() * 4 / ms / 1e9:.2f} TB/s of output)", flush=True)
