"""How far ahead of the GPU does the host run in the train loop (GPU box)?  Per step: wall time of _train_step, the part of
it spent blocked in the step's one host wait (the loss readback's event), and the rest = what the host needs to enqueue a step.
If enqueue time approaches the step time the GPU starts waiting for launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from robot_aware_control_amd import synthetic as syn
from robot_aware_control_amd.trainer import PredictionTrainer

dev = torch.device("cuda:0")
tr = PredictionTrainer(bench.namespace(dev))
tr.model.train()
data = syn.synth_video(seed=1, T=6, B=16)
data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}
blocked = [0.0]
real = torch.cuda.Event.synchronize


def timed_sync(self):
    t = time.perf_counter()
    real(self)
    blocked[0] += time.perf_counter() - t


torch.cuda.Event.synchronize = timed_sync
for _ in range(4):
    tr._train_step(data)
torch.cuda.synchronize()
n = 10
blocked[0] = 0.0
t0 = time.perf_counter()
for _ in range(n):
    tr._train_step(data)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"wall/step {(t2 - t0) / n * 1e3:.2f} ms; host in _train_step {(t1 - t0) / n * 1e3:.2f} ms of which blocked on the "
      f"readback {blocked[0] / n * 1e3:.2f} ms -> enqueue work {(t1 - t0 - blocked[0]) / n * 1e3:.2f} ms/step; "
      f"GPU still busy after the last return: {(t2 - t1) * 1e3:.2f} ms")
