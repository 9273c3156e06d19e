"""Micro-benchmark of rac_bn_bwd_reduce / rac_bn_bwd_apply on one layer shape (GPU box):
    python tools/bench_bn_reduce.py M C G        cold = a 1 GB buffer is streamed between launches (operands out of every cache)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robot_aware_control_amd import ops
from robot_aware_control_amd.ops import call, ptr, stream_ptr

M, C, G = [int(v) for v in sys.argv[1:4]]
dev = torch.device("cuda:0")
dy, x = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
aff = torch.randn(4, G, C, device=dev)
sums = torch.zeros(G, 2, C, device=dev, dtype=torch.float64)
dx = torch.empty_like(x)
big = torch.empty(1 << 28, device=dev)  # 1 GB
for cold in (False, True):
    for name, fn in (("reduce", lambda: call("rac_bn_bwd_reduce", ptr(dy), ptr(x), ptr(aff[0]), ptr(aff[1]), ptr(aff[2]), ptr(aff[3]),
                                             ptr(sums), M, C, G, stream_ptr())),
                     ("apply", lambda: call("rac_bn_bwd_apply", ptr(dy), ptr(x), ptr(aff[0]), ptr(aff[1]), ptr(aff[2]), ptr(aff[3]),
                                            ptr(sums), ptr(dx), None, None, M, C, G, None, stream_ptr()))):
        ts = []
        for it in range(12):
            if cold:
                big.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        print(f"M={M} C={C} G={G} {name} {'cold' if cold else 'warm'}: median {ts[len(ts)//2]:.1f} us  ({2 * M * C * 4 / ts[len(ts)//2] / 1e6:.2f} TB/s of dy + x)")
