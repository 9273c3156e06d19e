"""One frozen-model vgg conv on a map larger than a tile (the rows kernels), timed alone (GPU box).
usage: python tools/bench_rows.py B H W Cin Cout [iters]      e.g. 1000 16 16 256 256"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robot_aware_control_amd import ops

B, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = torch.device("cuda:0")
x = torch.randn(B, H, W, Cin, device=dev)
w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02).permute(0, 3, 1, 2)
scale, shift = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
ops.amax_for(x, True)
run = lambda: ops.conv_forward_split(x, None, w, None, act=ops.ACT_LEAKY, scale=scale, shift=shift, per_image=True)
for _ in range(2):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
flop = 2.0 * B * H * W * Cout * Cin * 9
print(f"rows conv B={B} {H}x{W} {Cin}->{Cout}: {ms:.3f} ms  {flop / ms / 1e9:.0f} TFLOP/s algorithmic ({flop / ms / 1e9 / 833.3 * 100:.1f} % of the split pipe)")
