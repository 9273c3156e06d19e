"""Micro-benchmark of rac_conv2d on the shapes that dominate the hot path (GPU box).
usage: python tools/bench_gemm.py [fwd|dgrad|wgrad] B [g] [k] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robot_aware_control_amd import ops

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
g = int(sys.argv[3]) if len(sys.argv) > 3 else 512
k = int(sys.argv[4]) if len(sys.argv) > 4 else 5
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda:0")
H = W = 8
x = torch.randn(B, H, W, g, device=dev)
h = torch.randn(B, H, W, g, device=dev)
w = (torch.randn(4 * g, k, k, 2 * g, device=dev) * 0.01).permute(0, 3, 1, 2)
dy = torch.randn(B, H, W, 4 * g, device=dev)
gw = torch.zeros_like(w)
w.grad = gw
flop = 2.0 * B * H * W * 4 * g * (2 * g * k * k)

def run():
    if mode == "fwd":
        return ops.conv_forward(x, h, w, None, want_slabs=True)
    if mode == "dgrad":
        return ops.conv_dgrad(dy, w, g, g)
    return ops.conv_wgrad_acc(dy, x, h, w)

for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"{mode} B={B} g={g} k={k} tile={os.environ.get('RAC_IGEMM_TILE','auto')} split={os.environ.get('RAC_SPLIT','auto')}: "
      f"{ms:.3f} ms  {flop / ms / 1e9:.1f} TFLOP/s  ({flop / ms / 1e9 / 157.3 * 100:.1f}% of fp32 MFMA peak)", flush=True)

def timeit(fn, label, nflop):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{label} B={B} g={g} k={k}: {ms:.3f} ms  {nflop / ms / 1e9:.1f} TFLOP/s algorithmic "
          f"({3 * nflop / ms / 1e9 / 2500 * 100:.1f}% of the fp16 MFMA peak at 3 products)", flush=True)


if os.environ.get("RAC_BENCH_SPLIT"):
    T = int(os.environ.get("RAC_BENCH_T", "1"))
    if mode == "fwd":
        ops.tag_amax(h, ops.amax_one(dev))
        timeit(lambda: ops.conv_forward_split(x, h, w, want_slabs=True), "fwd-split(f16x2, incl. absmax)", flop)
        ops.amax_for(x)
        timeit(lambda: ops.conv_forward_split(x, h, w, want_slabs=True), "fwd-split(f16x2, kernel only)", flop)
    elif mode == "dgrad":
        ops.amax_for(dy)
        timeit(lambda: ops.conv_dgrad_split(dy, w, g, g), "dgrad-split(f16x2 + slab reduce)", flop)
    else:
        items = [(dy, x, h)] * T
        timeit(lambda: ops._wgrad_split_batch(items, w), f"wgrad-split(f16x2) T={T}", T * flop)
