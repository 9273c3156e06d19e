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

if mode == "fwd" and os.environ.get("RAC_BENCH_SPLIT"):
    def run2():
        return ops.conv_forward_split(x, h, w, want_slabs=True)
    for _ in range(3):
        run2()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        run2()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"fwd-split(bf16x6) B={B} g={g} k={k}: {ms:.3f} ms  {flop / ms / 1e9:.1f} TFLOP/s effective  "
          f"({flop / ms / 1e9 / 157.3 * 100:.1f}% of the fp32 MFMA peak, {6 * flop / ms / 1e9 / 2500 * 100:.1f}% of bf16 peak x6)", flush=True)
    wl = int(os.environ.get('RAC_W_LAYOUT', '3'))
    pw = ops.split_parts({1: ops.chunk_major, 2: ops.frag_order, 3: ops.frag_order16}[wl](w))
    px, ph = (x, h) if wl >= 2 else (ops.split_parts(x), ops.split_parts(h))  # layouts 2 / 3 read fp32 activations
    aps = 0 if wl >= 2 else px.shape[1]
    import ctypes as C
    from robot_aware_control_amd._lib import ConvArgs, call, ptr, stream_ptr
    out = torch.empty((B, H, W, 4 * g), device=dev)
    args = ConvArgs(mode=0, B=B, H=H, W=W, ksize=k, Cin=2 * g, Cout=4 * g, act=0, split_k=1, accumulate=0, a_split=g, o_split=0, slab_stride=0,
                    a0=ptr(px), a1=ptr(ph), w=ptr(pw), out0=ptr(out), out1=None, bias=None, scale=None, shift=None, stats=None)
    for _ in range(2):
        call("rac_conv2d_fwd_split", C.byref(args), aps, aps, pw.shape[1], wl, stream_ptr())
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        call("rac_conv2d_fwd_split", C.byref(args), aps, aps, pw.shape[1], wl, stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"  kernel only: {ms:.3f} ms  {flop / ms / 1e9:.1f} TFLOP/s effective", flush=True)

if mode == "wgrad" and os.environ.get("RAC_BENCH_SPLIT"):
    def run3():
        return ops.conv_wgrad_split_acc(dy, x, h, w)
    for _ in range(3):
        run3()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        run3()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"wgrad-split(bf16x6) B={B} g={g} k={k}: {ms:.3f} ms  {flop / ms / 1e9:.1f} TFLOP/s effective (incl. transposes)", flush=True)

if mode == "wgrad" and os.environ.get("RAC_BENCH_SPLIT"):
    T = int(os.environ.get("RAC_BENCH_T", "1"))
    items = [(dy, x, h)] * T
    for _ in range(2):
        ops._wgrad_split_batch(items, w)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        ops._wgrad_split_batch(items, w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"wgrad-split(bf16x6) B={B} x T={T} g={g} k={k}: {ms:.3f} ms  {T * flop / ms / 1e9:.1f} TFLOP/s effective incl. transposes", flush=True)
