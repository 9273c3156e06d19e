"""Micro-benchmark of the 64 -> 4 output head's three forms (GPU box): python tools/bench_head.py [B] [H] [W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robot_aware_control_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
x = torch.randn(B, H, W, 64, device=dev)
w = (torch.randn(64, 4, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
b = torch.randn(4, device=dev) * 0.1
ops.tag_amax(x, ops.amax_of(x, per_image=True))
for name, mfma, direct in (("matrix pipe, roles swapped (rac_head_fwd_split)", True, True),
                           ("exact-fp32 FMAs (rac_head_fwd)", False, True), ("rows kernel, 32-column tile", False, False)):
    ops.HEAD_MFMA, ops.HEAD_DIRECT = mfma, direct
    with torch.no_grad():
        for _ in range(3):
            ops.ConvTHead.apply(x, w, b, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.ConvTHead.apply(x, w, b, True)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}: {ms * 1e3:.0f} us  ({x.numel() * 4 / ms / 1e9:.2f} TB/s of input)", flush=True)

# the head's data gradient (training: 80 images of 64 x 64): one streaming pass against the exact-fp32 implicit GEMM
Bt = 80
d = torch.randn(Bt, H, W, 4, device=dev)
dx = torch.empty(Bt, H, W, 64, device=dev)
forms = (("streaming pass (rac_head_dgrad)", lambda: ops.call("rac_head_dgrad", ops.ptr(d), ops.ptr(w), ops.ptr(dx), Bt, H, W, ops.stream_ptr())),
         ("exact-fp32 implicit GEMM", lambda: ops.conv_raw(ops.FWD, d, None, w, dx, B=Bt, H=H, W=W, ksize=3, Cin=4, Cout=64, a_split=4)))
for name, fn in forms:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"head dgrad, {name}: {ms * 1e3:.0f} us  ({dx.numel() * 4 / ms / 1e9:.2f} TB/s written)", flush=True)
