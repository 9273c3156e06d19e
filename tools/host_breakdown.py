"""Where the host's time goes in one train step (GPU box): wall-clock accumulators around the ABI calls, the tensor
allocations and the autograd nodes (no profiler: cProfile inflates the small calls several-fold).
    python tools/host_breakdown.py [tf|sched|deployed|deployed_sched]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from robot_aware_control_amd import _lib, ops, synthetic as syn
from robot_aware_control_amd.trainer import PredictionTrainer
mode = sys.argv[1] if len(sys.argv) > 1 else "deployed"
dev = torch.device("cuda:0")
kw = dict(g_dim=256, lstm_group_norm=True, image_height=48, model_use_future_robot_state=True) if mode.startswith("deployed") else {}
cf = bench.namespace(dev, **kw)
tr = PredictionTrainer(cf); tr.model.train()
tr.model.load_state_dict(syn.synth_state_dict(tr.model, seed=11))
tr.optimizer.overlap_next_forward = True
ut = [True, True, False, False, False, False] if mode.endswith("sched") else None
data = syn.synth_video(seed=1, T=6, B=16, H=cf.image_height, W=cf.image_width); data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}
for _ in range(4): tr._train_step(data, use_truth=ut)
torch.cuda.synchronize()
acc = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            e = acc.setdefault(key, [0, 0.0]); e[0] += 1; e[1] += time.perf_counter() - t
    setattr(obj, name, g)
wrap(_lib, "call", "abi call (ctypes + launch)"); ops.call = _lib.call
wrap(torch, "empty", "torch.empty"); wrap(torch, "empty_like", "torch.empty_like"); wrap(torch, "zeros", "torch.zeros")
wrap(torch, "cat", "torch.cat"); wrap(torch, "stack", "torch.stack")
wrap(ops, "amax_for", "amax_for"); wrap(ops, "weight_parts", "weight_parts"); wrap(ops, "plan_split_k", "plan_split_k")
wrap(_lib, "stream_ptr", "stream_ptr"); ops.stream_ptr = _lib.stream_ptr
wrap(torch.autograd, "backward", "autograd.backward (whole)")
n = 5
t0 = time.perf_counter()
for _ in range(n): tr._train_step(data, use_truth=ut)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"{mode}: host {(t1 - t0) / n * 1e3:.2f} ms/step")
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:32s} {c / n:8.0f} calls/step {t / n * 1e3:8.2f} ms/step  {t / c * 1e6:7.1f} us each")
