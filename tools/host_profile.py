"""Host-side cost of one train step (GPU box): cProfile of PredictionTrainer._train_step at cfg2."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from robot_aware_control_amd import synthetic as syn
from robot_aware_control_amd.trainer import PredictionTrainer
dev = torch.device("cuda:0")
cf = bench.namespace(dev)
tr = PredictionTrainer(cf); tr.model.train()
data = syn.synth_video(seed=1, T=6, B=16); data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}
for _ in range(3): tr._train_step(data)
torch.cuda.synchronize()
t0 = time.perf_counter(); 
for _ in range(3): tr._train_step(data)
torch.cuda.synchronize(); print("wall ms/step", (time.perf_counter() - t0) / 3 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tr._train_step(data)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
