"""Micro-benchmark of rac_bn_bwd_reduce / rac_bn_bwd_apply / rac_slab_reduce_stats on one layer shape (GPU box):
    python tools/bench_bn_reduce.py M C G [n_slabs]     20 back-to-back launches between two events (launch overhead out);
    cold = a 1 GB buffer is streamed between rounds (operands out of every cache).  A/B: RAC_BN_REDUCE_OLD=1, RAC_BN_REDUCE_COEF"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robot_aware_control_amd import ops
from robot_aware_control_amd.ops import call, ptr, stream_ptr

M, C, G = [int(v) for v in sys.argv[1:4]]
S = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda:0")
dy, x = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
slabs = torch.randn(S, M, C, device=dev)
aff = torch.randn(4, G, C, device=dev)
sums = torch.zeros(G, 2, C, device=dev, dtype=torch.float64)
dx = torch.empty_like(x)
slot = torch.zeros(1, device=dev, dtype=torch.int32)
big = torch.empty(1 << 28, device=dev)  # 1 GB
R = 20
fns = (("reduce", 2 * M * C * 4, lambda: call("rac_bn_bwd_reduce", ptr(dy), ptr(x), ptr(aff[0]), ptr(aff[1]), ptr(aff[2]), ptr(aff[3]),
                                              ptr(sums), M, C, G, stream_ptr())),
       ("apply", 3 * M * C * 4, lambda: call("rac_bn_bwd_apply", ptr(dy), ptr(x), ptr(aff[0]), ptr(aff[1]), ptr(aff[2]), ptr(aff[3]),
                                             ptr(sums), ptr(dx), None, None, M, C, G, ptr(slot), stream_ptr())),
       (f"slab_stats{S}", (S + 1) * M * C * 4, lambda: call("rac_slab_reduce_stats", ptr(slabs), S, M * C, ptr(dx), ptr(sums), M, C, G,
                                                           ptr(slot), stream_ptr())))
for name, nbytes, fn in fns:
    ts = []
    for it in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(R):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / R)
    ts = sorted(ts[2:])
    med = ts[len(ts) // 2]
    print(f"M={M} C={C} G={G} {name}: {med:.1f} us  ({nbytes / med / 1e6:.2f} TB/s)")
