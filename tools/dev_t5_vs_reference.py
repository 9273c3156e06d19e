import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_model as T
from oracle import svg_oracle as orc
from robot_aware_control_amd import synthetic as syn
g = np.load("/root/repo/tests/golden/train_t5_ra.npz")
cfg = orc.Cfg(g_dim=128, z_dim=16, batch_size=4, n_past=1, n_future=4, lr=1e-4, **T.FLAGSETS["ra"])
dev = torch.device("cuda:0")
tr = T.make_trainer(cfg, orc.make_weights(cfg, seed=6, randomize_bn_stats=False), dev)
data = syn.synth_video(seed=31, T=5, B=4)
queue = [e for pair in syn.synth_eps(seed=32, steps=4, B=4, z=16, h=8, w=8) for e in pair]
tr.model.eps_source = lambda shape: queue.pop(0)
tr.optimizer.step = lambda: None
losses = tr._train_step(data)
print({k: abs(losses[k] - float(g["loss_" + k])) / abs(float(g["loss_" + k])) for k in ("recon_loss", "robot_loss", "world_loss", "kld")})
grads = dict(tr.model.named_parameters())
pkeys = [k for k, _, kind in orc.param_spec(cfg) if not orc.is_buffer(kind)]
gn = np.array([grads[k].grad.double().norm().item() for k in pkeys])
r = np.abs(gn - g["grad_norms"]) / (np.abs(g["grad_norms"]) + 1e-30)
print("grad norm rel err: max %.2e median %.2e" % (r.max(), np.median(r)), pkeys[int(r.argmax())])
for name in ("prior0", "post1", "fp0", "fp_in", "head_mu", "dec", "enc"):
    key = str(g["gradkey_" + name]); ref = torch.from_numpy(g["grad_" + name])
    print(name, "%.2e" % T.rel(grads[key].grad[tuple(slice(0, n) for n in ref.shape)], ref))
