"""GPU-side duration of each train step right after a device synchronize (the timed region of bench.py starts like
that): shows the start-up transient.  usage: python tools/step_times.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from robot_aware_control_amd import synthetic as syn
from robot_aware_control_amd.trainer import PredictionTrainer

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if os.environ.get("NOGC"):
    import gc
    gc.disable()
dev = torch.device("cuda:0")
cf = bench.namespace(dev)
tr = PredictionTrainer(cf)
tr.model.train()
B, T = cf.batch_size, cf.n_past + cf.n_future
batches = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.synth_video(seed=100 + i, T=T, B=B).items()}
           for i in range(2)]
for i in range(6):
    tr._train_step(batches[i % 2])
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
host = []
t0 = time.perf_counter()
ev[0].record()
for i in range(K):
    tr._train_step(batches[i % 2])
    ev[i + 1].record()
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("GPU ms per step:", [round(ev[i].elapsed_time(ev[i + 1]), 1) for i in range(K)])
print("host returned from step i at ms:", [round(h * 1e3, 1) for h in host], "wall %.1f ms" % (wall * 1e3))
