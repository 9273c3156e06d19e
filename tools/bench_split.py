"""Split-K sweep of one split-precision conv shape (GPU box):  python tools/bench_split.py B H W Cin Cout k [fwd|dgrad]
Prints the time of conv + combine for forced K splits (RAC_SPLIT) next to the planner's choice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robot_aware_control_amd import ops

B, H, W, Cin, Cout, k = (int(v) for v in sys.argv[1:7])
mode = sys.argv[7] if len(sys.argv) > 7 else "fwd"
dev = torch.device("cuda:0")
x = torch.randn(B, H, W, Cin, device=dev)
dy = torch.randn(B, H, W, Cout, device=dev)
w = (torch.randn(Cout, k, k, Cin, device=dev) * 0.02).permute(0, 3, 1, 2)
ops.amax_for(x), ops.amax_for(dy)
flop = 2.0 * B * H * W * Cin * Cout * k * k
run = (lambda: ops.conv_forward_split(x, None, w, None)) if mode == "fwd" else (lambda: ops.conv_dgrad_split(dy, w, Cin, 0))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for s in ("", "1", "2", "3", "4", "6", "8"):
    if s:
        os.environ["RAC_SPLIT"] = s
    else:
        os.environ.pop("RAC_SPLIT", None)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{mode} M={B * H * W} N={Cout if mode == 'fwd' else Cin} K={(Cin if mode == 'fwd' else Cout) * k * k} split={s or 'planner'}: "
          f"{us:.1f} us  {flop / us / 1e6:.0f} TF ({flop / us / 1e6 / 833.3:.2f})", flush=True)
