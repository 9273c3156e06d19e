"""Host-side cost of one train step (GPU box): enqueue time and cProfile of PredictionTrainer._train_step.
    python tools/host_profile.py [tf|sched|deployed|deployed_sched]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from robot_aware_control_amd import synthetic as syn
from robot_aware_control_amd.trainer import PredictionTrainer
mode = sys.argv[1] if len(sys.argv) > 1 else "tf"
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
blocked = [0.0]
real = torch.cuda.Event.synchronize
def timed_sync(self):
    t = time.perf_counter(); real(self); blocked[0] += time.perf_counter() - t
torch.cuda.Event.synchronize = timed_sync
n = 10
t0 = time.perf_counter()
for _ in range(n): tr._train_step(data, use_truth=ut)
t1 = time.perf_counter()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{mode}: wall/step {(t2 - t0) / n * 1e3:.2f} ms; host in _train_step {(t1 - t0) / n * 1e3:.2f} ms of which blocked on the readback "
      f"{blocked[0] / n * 1e3:.2f} ms -> enqueue work {(t1 - t0 - blocked[0]) / n * 1e3:.2f} ms/step; GPU busy after the last return {(t2 - t1) * 1e3:.2f} ms")
torch.cuda.Event.synchronize = real
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tr._train_step(data, use_truth=ut)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
