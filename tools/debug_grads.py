"""Debug helper (GPU box): per-parameter gradient error of one train step vs the CPU oracle."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import svg_oracle as orc
from robot_aware_control_amd import synthetic as syn
from robot_aware_control_amd.trainer import PredictionTrainer

tag = sys.argv[1] if len(sys.argv) > 1 else "vanilla"
g_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 2
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
FL = {"vanilla": dict(model_use_mask=False, model_use_future_mask=False, model_use_robot_state=False, reconstruction_loss="l1"),
      "ra": dict(model_use_mask=True, model_use_future_mask=True, model_use_robot_state=True, reconstruction_loss="dontcare_l1")}
cfg = orc.Cfg(g_dim=g_dim, z_dim=16, batch_size=B, n_past=1, n_future=nf, lr=1e-4, **FL[tag])
dev = torch.device("cuda:0")
d = dict(cfg.__dict__); d.update(device=dev, experiment="train_robonet", load_movement_info=False, movement_weight=1.0, scheduled_sampling=False, scheduled_sampling_k=4000, model="svg", optimizer="adam", seed=0, wandb=False, log_dir="/tmp/x", dynamics_model_ckpt=None, ddp_bucket_mb=64)
sd = orc.make_weights(cfg, seed=1, randomize_bn_stats=False)
data = syn.synth_video(seed=20, T=nf+1, B=B)
eps = syn.synth_eps(seed=40, steps=nf, B=B, z=16, h=8, w=8)
ts = orc.TrainState.create(cfg, sd)
ref = orc.train_step(ts, data, eps, None, do_update=False)
tr = PredictionTrainer(argparse.Namespace(**d))
tr.model.load_state_dict({k: v.clone() for k, v in sd.items()}); tr.model.train()
q = [e for p in eps for e in p]
tr.model.eps_source = lambda s: q.pop(0)
tr.optimizer.step = lambda: None
got = tr._train_step(data)
print("losses", {k: (got[k], ref[k]) for k in ref})
grads = dict(tr.model.named_parameters())
rows = []
for k in ts.param_keys:
    a, b = grads[k].grad.double().cpu(), ts.sd[k].grad.double()
    rows.append((float((a-b).norm()/(b.norm()+1e-30)), k, float(b.norm())))
for e, k, n in rows:
    print(f"{e:10.3e}  {n:10.3e}  {k}")

# ---- layer-wise gradient w.r.t. every vgg block output ----
if os.environ.get("RAC_TRACE"):
    from robot_aware_control_amd import model as M
    orc.TRACE = []
    ts = orc.TrainState.create(cfg, sd)
    orc.train_step(ts, data, eps, None, do_update=False)
    ref_tr = orc.TRACE; orc.TRACE = None
    names = {}
    for n, m in tr.model.named_modules():
        if isinstance(m, M._VggLayer): names[id(m)] = n
    got_tr = []
    orig = M._VggLayer.forward
    def fwd(self, x0, x1=None, n_updates=1):
        y = orig(self, x0, x1, n_updates)
        rec = [names[id(self)], y.detach(), None]
        y.register_hook(lambda g, rec=rec: rec.__setitem__(2, g.detach().clone()))
        got_tr.append(rec)
        return y
    M._VggLayer.forward = fwd
    q.extend(e for p in eps for e in p)
    tr._train_step(data)
    # the oracle runs the encoder twice per step (C4): merge by summing the two passes' grads
    merged = {}
    order = []
    ri = 0
    step_len_ref = 10 * 2 + 9
    for si in range(nf):
        blk = ref_tr[si * step_len_ref:(si + 1) * step_len_ref]
        enc1, enc2, dec = blk[:10], blk[10:20], blk[20:]
        for (n1, t1), (n2, t2) in zip(enc1, enc2):
            g = t1.grad + (t2.grad if t2.grad is not None else 0)
            order.append((f"s{si}.{n1}", t1.detach(), g))
        for n1, t1 in dec:
            order.append((f"s{si}.{n1}", t1.detach(), t1.grad))
    for (rn, ry, rg), (gn, gy, gg) in zip(order, got_tr):
        ya = gy.permute(0, 3, 1, 2).cpu().double(); ga = gg.permute(0, 3, 1, 2).cpu().double()
        ey = float((ya - ry.double()).norm() / ry.double().norm())
        eg = float((ga - rg.double()).norm() / rg.double().norm())
        flips = int(((ya > 0) != (ry.double() > 0)).sum())
        print(f"{rn:34s} {gn:24s} fwd_err {ey:9.2e} grad_err {eg:9.2e} slope_flips {flips} of {ya.numel()}")
