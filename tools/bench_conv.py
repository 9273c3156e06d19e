"""Micro-benchmark of one split-precision conv shape (GPU box):
    python tools/bench_conv.py B H W C0 C1 Cout k [iters]        forward with the folded-BatchNorm + LeakyReLU epilogue
prints ms and algorithmic TFLOP/s for the forward conv and for the data gradient (the same kernel on dy)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robot_aware_control_amd import ops

B, H, W, C0, C1, Cout, k = [int(v) for v in sys.argv[1:8]]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
dev = torch.device("cuda:0")
Cin = C0 + C1
x0 = torch.randn(B, H, W, C0, device=dev)
x1 = torch.randn(B, H, W, C1, device=dev) if C1 else None
w = (torch.randn(Cout, k, k, Cin, device=dev) * 0.02).permute(0, 3, 1, 2)
scale, shift = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
dy = torch.randn(B, H, W, Cout, device=dev)
flop = 2.0 * B * H * W * Cout * Cin * k * k
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(fn, label):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{label} B={B} {H}x{W} {C0}+{C1}->{Cout} k{k}: {ms:.3f} ms  {flop / ms / 1e9:.0f} TFLOP/s "
          f"({flop / ms / 1e9 / 833.3:.2f} of peak)", flush=True)


timeit(lambda: ops.conv_forward_split(x0, x1, w, None, act=ops.ACT_LEAKY, scale=scale, shift=shift), "fwd  ")
if os.environ.get("BENCH_DGRAD", "1") == "1":
    timeit(lambda: ops.conv_dgrad_split(dy, w, C0, C1), "dgrad")
