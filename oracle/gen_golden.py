"""TEST INFRASTRUCTURE ONLY -- golden-vector generator.

Runs ONLY in the build container: imports the real reference from
/root/reference (absent third-party deps replaced by MagicMock modules, the
recipe of SURVEY.md Appendix A), loads the name-keyed synthetic weights of
`oracle.svg_oracle.make_weights`, runs the reference hot path on the synthetic
inputs of `robot_aware_control_amd.synthetic` and writes small `.npz` fixtures
to tests/golden/.  The reference itself never travels; only these vectors do.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden.py
"""
import argparse
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# our own modules are loaded by path so that the name `src` stays the reference's
orc = _load("svg_oracle", os.path.join(REPO, "oracle", "svg_oracle.py"))
syn = _load("rac_synthetic", os.path.join(REPO, "robot_aware_control_amd", "synthetic.py"))

MISSING = {"skimage", "torchvision", "h5py", "imageio", "wandb", "colorlog", "ipdb", "cv2", "mujoco_py", "gym",
           "rospy", "actionlib", "eef_control", "cv_bridge", "sensor_msgs", "pupil_apriltags", "mujoco", "glfw",
           "franka_ik_service", "sklearn"}


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in MISSING:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__, m.__spec__, m.__name__ = [], spec, spec.name
        return m

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _StubFinder())
sys.path.insert(0, REF)
torch.set_num_threads(8)

# ---- eps injection: Tensor.normal_() pops from this queue when it is non-empty ----
_EPS = []
_orig_normal = torch.Tensor.normal_


def _normal_patch(self, mean=0.0, std=1.0, *, generator=None):
    if _EPS:
        e = _EPS.pop(0)
        assert tuple(e.shape) == tuple(self.shape), (e.shape, self.shape)
        self.copy_(e)
        return self
    return _orig_normal(self, mean, std, generator=generator)


torch.Tensor.normal_ = _normal_patch

from src.prediction.models.dynamics import SVGConvModel  # noqa: E402
import src.prediction.losses as ref_losses  # noqa: E402
from src.utils.state import DemoGoalState, State  # noqa: E402

FLAGSETS = {
    "vanilla": dict(model_use_mask=False, model_use_future_mask=False, model_use_robot_state=False,
                    reconstruction_loss="l1"),
    "ra": dict(model_use_mask=True, model_use_future_mask=True, model_use_robot_state=True,
               reconstruction_loss="dontcare_l1"),
}


def ns_for(cfg: "orc.Cfg", **extra):
    d = dict(cfg.__dict__)
    d.update(device=torch.device("cpu"), debug_cem=False, log_dir="/tmp/rac_golden", img_cost_threshold=None,
             img_cost_world_norm=True, experiment="train_robonet", robot_joint_dim=5, multiview=False,
             load_movement_info=False, scheduled_sampling=False, model="svg", optimizer="adam")
    d.update(extra)
    return argparse.Namespace(**d)


def ref_model(cfg, sd, train=False):
    m = SVGConvModel(ns_for(cfg))
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m.train(train)
    return m


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KB)")


def step_inputs(cfg, data, i):
    """The argument packing of trainer.py:352-404 for time index i (ground-truth token)."""
    x, m, s, a = data["images"], data["masks"], data["states"], data["actions"]
    x_j, x_i, m_j, m_i = x[i - 1], x[i], m[i - 1], m[i]
    if "dontcare" in cfg.reconstruction_loss or cfg.black_robot_input:
        x_j, x_i = orc.zero_robot_region(m_j, x_j), orc.zero_robot_region(m_i, x_i)
    m_in = torch.cat([m_j, m_i], 1) if cfg.model_use_future_mask else m_j
    m_next = m_i.repeat(1, 2, 1, 1) if cfg.model_use_future_mask else m_i
    return x_j, m_in, s[i - 1], a[i - 1], x_i, m_next, s[i]


def gen_forward():
    """F2: SVGConvModel.forward, eval + train mode, two consecutive steps."""
    for tag, flags in FLAGSETS.items():
        cfg = orc.Cfg(g_dim=64, z_dim=16, batch_size=2, **flags)
        sd = orc.make_weights(cfg, seed=7)
        data = syn.synth_video(seed=3, T=3, B=2)
        eps = syn.synth_eps(seed=5, steps=2, B=2, z=16, h=8, w=8)
        for mode in ("eval", "train"):
            m = ref_model(cfg, sd, train=(mode == "train"))
            m.init_hidden(2)
            out = {}
            with torch.no_grad():
                # step 1: posterior path (next frame given), eps injected prior-then-posterior
                x_j, m_in, r, a, x_i, m_next, r_i = step_inputs(cfg, data, 1)
                _EPS.extend([eps[0][0], eps[0][1]])
                o = m(x_j, m_in, r, None, a, x_i, m_next, r_i, None, None)
                assert not _EPS
                for nm, v in zip(("x_pred", "skip", "mu", "logvar", "mu_p", "logvar_p"), o):
                    if nm == "skip":
                        for k, sk in enumerate(v):
                            out[f"s1_skip{k}_sum"] = sk.double().sum()
                            out[f"s1_skip{k}_abs"] = sk.double().abs().sum()
                        out["s1_skip3"] = v[3]
                    else:
                        out[f"s1_{nm}"] = v
                # step 2: prior only, sample_mean (the CEM call form, trajectory_sampler.py:148)
                x_j, m_in, r, a, _, _, _ = step_inputs(cfg, data, 2)
                _EPS.extend([eps[1][0]])
                o = m.forward(x_j, m_in, r, None, a, sample_mean=True)
                assert not _EPS
                out["s2_x_pred"], out["s2_mu_p"], out["s2_logvar_p"] = o[0], o[4], o[5]
                assert o[2] is None and o[3] is None
            if mode == "train":
                st = m.state_dict()
                for k in ("encoder.c1.0.main.1", "encoder.c4.2.main.1", "decoder.upc5.0.main.1"):
                    out[k + ".running_mean"] = st[k + ".running_mean"].clone()
                    out[k + ".running_var"] = st[k + ".running_var"]
                    out[k + ".num_batches_tracked"] = st[k + ".num_batches_tracked"]
            save(f"fwd_{mode}_{tag}", **out)


def gen_heatmap():
    """M4 with `--model_use_heatmap True --model_use_future_heatmap True` (dynamics.py:476-487,578-582): the heatmap
    planes sit between the image and the mask in the encoder input."""
    cfg = orc.Cfg(g_dim=32, z_dim=8, batch_size=2, model_use_heatmap=True, model_use_future_heatmap=True,
                  **FLAGSETS["ra"])
    sd = orc.make_weights(cfg, seed=13)
    data = syn.synth_video(seed=6, T=2, B=2)
    hm = torch.from_numpy(np.random.Generator(np.random.Philox(key=[6, 99])).random((2, 2, 1, 64, 64), dtype=np.float32))
    eps = syn.synth_eps(seed=7, steps=1, B=2, z=8, h=8, w=8)
    m = ref_model(cfg, sd, train=False)
    m.init_hidden(2)
    with torch.no_grad():
        x_j, m_in, r, a, x_i, m_next, r_i = step_inputs(cfg, data, 1)
        hm_in, hm_next = torch.cat([hm[0], hm[1]], 1), hm[1].repeat(1, 2, 1, 1)
        _EPS.extend([eps[0][0], eps[0][1]])
        o = m(x_j, m_in, r, hm_in, a, x_i, m_next, r_i, hm_next, None)
        assert not _EPS
    assert m.encoder.c1[0].main[0].weight.shape[1] == 7
    save("fwd_heatmap", x_pred=o[0], mu=o[2], mu_p=o[4], logvar_p=o[5], heatmaps=hm)


def gen_shape_pin():
    """F6: 48x64 input -> 6x8 latent."""
    cfg = orc.Cfg(g_dim=32, z_dim=8, batch_size=1, image_height=48, image_width=64, **FLAGSETS["vanilla"])
    sd = orc.make_weights(cfg, seed=2)
    m = ref_model(cfg, sd)
    m.init_hidden(1)
    data = syn.synth_video(seed=4, T=2, B=1, H=48, W=64)
    with torch.no_grad():
        o = m.forward(data["images"][0], None, None, None, data["actions"][0], sample_mean=True)
    _EPS.clear()
    save("fwd_48x64", x_pred=o[0], mu_p=o[4])


def gen_losses():
    """F3: loss values and input gradients."""
    g = np.random.Generator(np.random.Philox(key=[11, 0]))
    pred = torch.from_numpy(g.random((3, 3, 16, 16), dtype=np.float32)).requires_grad_(True)
    target = torch.from_numpy(g.random((3, 3, 16, 16), dtype=np.float32))
    mask = torch.from_numpy((g.random((3, 1, 16, 16), dtype=np.float32) < 0.3).astype(np.float32))
    bw = torch.tensor([1.0, 2.5, 0.5])
    out = {"pred": pred.detach(), "target": target, "mask": mask, "bw": bw}

    def rec(name, fn):
        pred.grad = None
        v = fn()
        v.backward()
        out[name], out[name + "_grad"] = v.detach(), pred.grad.clone()

    rec("l1", lambda: ref_losses.l1_criterion(pred, target))
    rec("l1_bw", lambda: ref_losses.l1_criterion(pred, target, bw))
    rec("mse", lambda: ref_losses.mse_criterion(pred, target))
    rec("dc_l1_w0", lambda: ref_losses.dontcare_l1_criterion(pred, target, mask, 0))
    rec("dc_l1_w05", lambda: ref_losses.dontcare_l1_criterion(pred, target, mask, 0.5))
    rec("dc_l1_w0_bw", lambda: ref_losses.dontcare_l1_criterion(pred, target, mask, 0, bw))
    rec("dc_mse_w05", lambda: ref_losses.dontcare_mse_criterion(pred, target, mask, 0.5))
    with torch.no_grad():
        out["robot_mse"] = ref_losses.robot_mse_criterion(pred, target, mask)
        out["world_mse"] = ref_losses.world_mse_criterion(pred, target, mask)
    mu1, lv1, mu2, lv2 = [torch.from_numpy(g.standard_normal((3, 4, 8, 8), dtype=np.float32) * np.float32(0.7))
                          .requires_grad_(True) for _ in range(4)]
    kl = ref_losses.kl_criterion(mu1, lv1, mu2, lv2, 3)
    kl.backward()
    out.update(kl=kl.detach(), mu1=mu1.detach(), lv1=lv1.detach(), mu2=mu2.detach(), lv2=lv2.detach(),
               kl_gmu1=mu1.grad, kl_glv1=lv1.grad, kl_gmu2=mu2.grad, kl_glv2=lv2.grad)
    # CEM costs
    cost_cfg = argparse.Namespace(robot_cost_weight=0, world_cost_weight=1, reward_type="dense",
                                  img_cost_threshold=None, img_cost_world_norm=True)
    curr = torch.from_numpy(g.random((4, 3, 16, 16), dtype=np.float32))
    goal = torch.from_numpy(g.random((3, 16, 16), dtype=np.float32))
    cm = torch.from_numpy((g.random((4, 1, 16, 16), dtype=np.float32) < 0.3))
    gm = torch.from_numpy((g.random((1, 16, 16), dtype=np.float32) < 0.3))
    out.update(c_curr=curr, c_goal=goal, c_cmask=cm, c_gmask=gm)
    out["cost_l2"] = ref_losses.RobotWorldCost(cost_cfg)(State(img=curr), State(img=goal))
    cost_cfg.reward_type = "dontcare"
    out["cost_dontcare"] = ref_losses.RobotWorldCost(cost_cfg)(State(img=curr, mask=cm), State(img=goal, mask=gm))
    save("losses", **out)


def gen_train():
    """F4: PredictionTrainer._train_step x3 at cfg1 (+ one scheduled-sampling run)."""
    from src.prediction.trainer import PredictionTrainer
    for tag, flags in FLAGSETS.items():
        for sched in ((False,) if tag == "vanilla" else (False, True)):
            cfg = orc.Cfg(g_dim=64, z_dim=16, batch_size=2, n_past=1, n_future=2, lr=1e-4, **flags)
            sd = orc.make_weights(cfg, seed=1, randomize_bn_stats=False)
            ns = ns_for(cfg, wandb=False, jobname="g", wandb_project="x", wandb_entity="x", wandb_group=None,
                        wandb_job_type=None, img_augmentation=False, seed=0, scheduled_sampling=sched,
                        scheduled_sampling_k=4000, learned_robot_model=False)
            tr = PredictionTrainer(ns)
            tr.model.load_state_dict({k: v.clone() for k, v in sd.items()})
            tr.model.train()
            tr._step = 0
            flips = [False, True, False]  # consumed once per step at i == 2
            if sched:
                tr._use_true_token = lambda: flips.pop(0)
            out = {"flips": np.array([False, True, False])}
            for step in range(3):
                data = syn.synth_video(seed=20 + step, T=3, B=2)
                eps = syn.synth_eps(seed=40 + step, steps=2, B=2, z=16, h=8, w=8)
                for e in eps:
                    _EPS.extend(e)
                losses = tr._train_step(data)
                assert not _EPS
                for k, v in losses.items():
                    out[f"step{step}_{k}"] = v
                st = tr.model.state_dict()
                keys = [k for k, _, kind in orc.param_spec(cfg) if kind != "bn_nbt"]
                out[f"step{step}_norms"] = np.array([st[k].double().norm().item() for k in keys])
                if step == 0:
                    grads = dict(tr.model.named_parameters())
                    pk = [k for k, _, kind in orc.param_spec(cfg) if not orc.is_buffer(kind)]
                    out["step0_grad_norms"] = np.array([grads[k].grad.double().norm().item() for k in pk])
                    out["step0_grad_slice_enc"] = grads["encoder.c1.0.main.0.weight"].grad.clone()
                    out["step0_grad_slice_lstm"] = grads["prior.lstm.1.gates.weight"].grad[:4, :8].clone()
                out[f"step{step}_nbt_enc"] = st["encoder.c1.0.main.1.num_batches_tracked"].clone()
                out[f"step{step}_rm_enc"] = st["encoder.c1.1.main.1.running_mean"].clone()
                out[f"step{step}_rv_enc"] = st["encoder.c1.1.main.1.running_var"].clone()
                out[f"step{step}_rm_dec"] = st["decoder.upc4.1.main.1.running_mean"].clone()
                out[f"step{step}_w_slice"] = st["frame_predictor.lstm.0.gates.weight"][:2, :3].clone()
            save(f"train_cfg1_{tag}" + ("_sched" if sched else ""), **out)


def gen_train_t5():
    """F4b: one PredictionTrainer._train_step of a FIVE-frame window (n_future 4: the shape of BASELINE configs[1]'s
    recurrence) at g 128 / z 16, batch 4 -- the size at which the HIP path takes its hand-scheduled recurrent core
    (layer-major order, thin convs batched over the steps): losses, every parameter's gradient norm, gradient slices of
    each ConvLSTM chain, the decoder and the encoder, and the weights' norms behind the optimiser step."""
    from src.prediction.trainer import PredictionTrainer
    cfg = orc.Cfg(g_dim=128, z_dim=16, batch_size=4, n_past=1, n_future=4, lr=1e-4, **FLAGSETS["ra"])
    sd = orc.make_weights(cfg, seed=6, randomize_bn_stats=False)
    ns = ns_for(cfg, wandb=False, jobname="g", wandb_project="x", wandb_entity="x", wandb_group=None,
                wandb_job_type=None, img_augmentation=False, seed=0, scheduled_sampling=False,
                scheduled_sampling_k=4000, learned_robot_model=False)
    tr = PredictionTrainer(ns)
    tr.model.load_state_dict({k: v.clone() for k, v in sd.items()})
    tr.model.train()
    tr._step = 0
    data = syn.synth_video(seed=31, T=5, B=4)
    eps = syn.synth_eps(seed=32, steps=4, B=4, z=16, h=8, w=8)
    for e in eps:
        _EPS.extend(e)
    losses = tr._train_step(data)
    assert not _EPS
    out = {f"loss_{k}": v for k, v in losses.items()}
    grads = dict(tr.model.named_parameters())
    pk = [k for k, _, kind in orc.param_spec(cfg) if not orc.is_buffer(kind)]
    out["grad_norms"] = np.array([grads[k].grad.double().norm().item() for k in pk])
    for name, key, sl in (("prior0", "prior.lstm.0.gates.weight", (slice(0, 4), slice(0, 8))),
                          ("post1", "posterior.lstm.1.gates.weight", (slice(0, 4), slice(0, 8))),
                          ("fp0", "frame_predictor.lstm.0.gates.weight", (slice(0, 4), slice(0, 8))),
                          ("fp_in", "frame_pred_input_conv.weight", (slice(0, 8), slice(None))),
                          ("head_mu", "posterior.mu_net.weight", (slice(0, 4), slice(0, 16))),
                          ("dec", "decoder.upc4.1.main.0.weight", (slice(0, 4), slice(None))),
                          ("enc", "encoder.c1.0.main.0.weight", (slice(None), slice(None)))):
        if key in grads:
            out[f"grad_{name}"] = grads[key].grad[sl].clone()
            out[f"gradkey_{name}"] = np.array(key)
    st = tr.model.state_dict()
    keys = [k for k, _, kind in orc.param_spec(cfg) if kind != "bn_nbt"]
    out["norms_after"] = np.array([st[k].double().norm().item() for k in keys])
    save("train_t5_ra", **out)


SWEEP = {
    # future robot state + black_robot_input + dontcare_mse with a robot pixel weight, two context frames, skip
    # connections frozen after the context (last_frame_skip False)
    "a": dict(model_use_mask=True, model_use_future_mask=False, model_use_robot_state=True,
              model_use_future_robot_state=True, black_robot_input=True, reconstruction_loss="dontcare_mse",
              robot_pixel_weight=0.3, last_frame_skip=False, n_past=2, n_future=2),
    # plain mse on a non-square 48x64 frame, three samples
    "b": dict(reconstruction_loss="mse", image_height=48, image_width=64, batch_size=3),
}


def gen_sweep():
    """One PredictionTrainer._train_step of the real reference for flag / shape combinations outside cfg1."""
    from src.prediction.trainer import PredictionTrainer
    for tag, flags in SWEEP.items():
        kw = dict(g_dim=32, z_dim=8, batch_size=2, n_past=1, n_future=2, lr=1e-4)
        kw.update(flags)
        cfg = orc.Cfg(**kw)
        B, T, H, W = cfg.batch_size, cfg.n_past + cfg.n_future, cfg.image_height, cfg.image_width
        sd = orc.make_weights(cfg, seed=6, randomize_bn_stats=False)
        ns = ns_for(cfg, wandb=False, jobname="g", wandb_project="x", wandb_entity="x", wandb_group=None,
                    wandb_job_type=None, img_augmentation=False, seed=0, scheduled_sampling_k=4000,
                    learned_robot_model=False)
        tr = PredictionTrainer(ns)
        tr.model.load_state_dict({k: v.clone() for k, v in sd.items()})
        tr.model.train()
        tr._step = 0
        for e in syn.synth_eps(seed=32, steps=T - 1, B=B, z=cfg.z_dim, h=H // 8, w=W // 8):
            _EPS.extend(e)
        losses = tr._train_step(syn.synth_video(seed=31, T=T, B=B, H=H, W=W))
        assert not _EPS
        out = {f"train_{k}": v for k, v in losses.items()}
        grads = dict(tr.model.named_parameters())
        pk = [k for k, _, kind in orc.param_spec(cfg) if not orc.is_buffer(kind)]
        out["train_grad_norms"] = np.array([grads[k].grad.double().norm().item() for k in pk])
        st = tr.model.state_dict()
        out["rm_enc"] = st["encoder.c1.1.main.1.running_mean"].clone()
        save(f"sweep_{tag}", **out)


CEM_GAIN = 200.0
CEM_SEED = {"vanilla": 5, "ra": 4}


def gen_groupnorm():
    """NormConvLSTMCell (--lstm_group_norm True): two eval forwards + one train step."""
    from src.prediction.trainer import PredictionTrainer
    cfg = orc.Cfg(g_dim=64, z_dim=16, batch_size=2, n_past=1, n_future=2, lr=1e-4, lstm_group_norm=True,
                  **FLAGSETS["ra"])
    sd = orc.make_weights(cfg, seed=11)
    data = syn.synth_video(seed=3, T=3, B=2)
    eps = syn.synth_eps(seed=5, steps=2, B=2, z=16, h=8, w=8)
    m = ref_model(cfg, sd)
    m.init_hidden(2)
    out = {}
    with torch.no_grad():
        x_j, m_in, r, a, x_i, m_next, r_i = step_inputs(cfg, data, 1)
        _EPS.extend([eps[0][0], eps[0][1]])
        o = m(x_j, m_in, r, None, a, x_i, m_next, r_i, None, None)
        out["s1_x_pred"], out["s1_mu"], out["s1_logvar_p"] = o[0], o[2], o[5]
        x_j, m_in, r, a, _, _, _ = step_inputs(cfg, data, 2)
        _EPS.extend([eps[1][0]])
        o = m.forward(x_j, m_in, r, None, a, sample_mean=True)
        out["s2_x_pred"], out["s2_mu_p"] = o[0], o[4]
    ns = ns_for(cfg, wandb=False, jobname="g", wandb_project="x", wandb_entity="x", wandb_group=None,
                wandb_job_type=None, img_augmentation=False, seed=0, scheduled_sampling_k=4000,
                learned_robot_model=False)
    tr = PredictionTrainer(ns)
    sd2 = orc.make_weights(cfg, seed=12, randomize_bn_stats=False)
    tr.model.load_state_dict({k: v.clone() for k, v in sd2.items()})
    tr.model.train()
    tr._step = 0
    for e in syn.synth_eps(seed=40, steps=2, B=2, z=16, h=8, w=8):
        _EPS.extend(e)
    losses = tr._train_step(syn.synth_video(seed=20, T=3, B=2))
    for k, v in losses.items():
        out[f"train_{k}"] = v
    grads = dict(tr.model.named_parameters())
    pk = [k for k, _, kind in orc.param_spec(cfg) if not orc.is_buffer(kind)]
    out["train_grad_norms"] = np.array([grads[k].grad.double().norm().item() for k in pk])
    save("groupnorm_ra", **out)


def gen_eval():
    """_eval_step (1-step and autoregressive) + raw psnr/ssim vectors."""
    from src.prediction.trainer import PredictionTrainer
    from src.utils.metrics import psnr as ref_psnr, ssim as ref_ssim
    cfg = orc.Cfg(g_dim=64, z_dim=16, batch_size=2, n_past=1, n_future=2, **FLAGSETS["ra"])
    sd = orc.make_weights(cfg, seed=7)
    ns = ns_for(cfg, wandb=False, jobname="g", wandb_project="x", wandb_entity="x", wandb_group=None,
                wandb_job_type=None, img_augmentation=False, seed=0, scheduled_sampling_k=4000,
                learned_robot_model=False, n_eval=4, test_batch_size=2)
    tr = PredictionTrainer(ns)
    tr.model.load_state_dict({k: v.clone() for k, v in sd.items()})
    tr.model.eval()
    data = syn.synth_video(seed=31, T=4, B=2)
    data["pred_masks"] = data["masks"]
    out = {}
    for autoreg in (False, True):
        for e in syn.synth_eps(seed=50, steps=3, B=2, z=16, h=8, w=8):
            _EPS.extend(e)
        losses = tr._eval_step(data, autoregressive=autoreg)
        assert not _EPS
        for k, v in losses.items():
            out[f"{'ar' if autoreg else 'one'}:{k}"] = v
    g = np.random.Generator(np.random.Philox(key=[13, 0]))
    a = torch.from_numpy(g.random((2, 3, 48, 64), dtype=np.float32))
    b = (a + torch.from_numpy(g.standard_normal((2, 3, 48, 64), dtype=np.float32)) * 0.1).clamp(0, 1)
    out["m_a"], out["m_b"] = a, b
    out["m_psnr"] = ref_psnr(a, b)
    out["m_ssim"] = ref_ssim(a, b)
    save("eval_ra", **out)


class _FakeRobotModel:
    def __init__(self, states, masks):
        self.states, self.masks = states, masks

    def predict_batch(self, start_data, thick=True):
        return self.states.clone(), self.masks.clone()


def gen_cem():
    """F5: generate_model_rollouts + get_action traces, action-amplified weights."""
    from src.cem.cem import CEMPolicy
    for tag, flags in FLAGSETS.items():
        ra = tag == "ra"
        cfg = orc.Cfg(g_dim=64, z_dim=16, batch_size=2, candidates_batch_size=5, sample_mean=True,
                      reward_type="dontcare" if ra else "dense", topk=3, **flags)
        sd = orc.make_weights(cfg, seed=9, action_gain=CEM_GAIN)
        model = ref_model(cfg, sd)
        off = dict(model_use_mask=False, model_use_robot_state=False, black_robot_input=False,
                   reward_type="dense", reconstruction_loss="l1")
        N, T = 12, 4
        prob = syn.synth_cem_problem(seed=CEM_SEED[tag], N=N + 1, T=T, with_robot=ra, goal_blend=0.15)
        pol = CEMPolicy(ns_for(cfg, **off), model, horizon=T + 1, opt_iter=2, action_candidates=N, topk=3,
                        init_std=0.03)
        ns_on = ns_for(cfg)
        pol.cfg = pol.traj_sampler.cfg = ns_on
        pol.traj_sampler.cost = ref_losses.RobotWorldCost(ns_on)
        start = State(img=prob["start_img"], state=np.zeros(5, np.float32), qpos=np.zeros(5, np.float32))
        goal = DemoGoalState(imgs=prob["goal_imgs"], masks=prob["goal_masks"])
        out = {}
        # (a) plain rollouts with an opt_traj appended as candidate N+1 (trajectory_sampler.py:62-68)
        if ra:
            pol.traj_sampler.robot_model = _FakeRobotModel(prob["states"], prob["masks"])
        acts = prob["actions"][:N]
        opt_traj = prob["actions"][N, :, :2].clone()
        ro = pol.traj_sampler.generate_model_rollouts(acts.clone(), start, goal, opt_traj=opt_traj)
        out["ro_sum_cost"], out["ro_optimal_sum_cost"] = ro["sum_cost"], ro["optimal_sum_cost"]
        srt = np.sort(ro["sum_cost"])[::-1]
        out["ro_gap_k"] = (srt[2] - srt[3]) / abs(srt[2])
        # (b) get_action trace: record what Normal.sample consumed and every refit
        if ra:
            pol.traj_sampler.robot_model = _FakeRobotModel(prob["states"][:, :N], prob["masks"][:, :N])
        trace = []
        orig = pol._get_rollouts

        def spy(act_seq, start_, goal_, opt_traj=None, plot=False):
            r = orig(act_seq, start_, goal_, opt_traj, plot)
            trace.append((act_seq.clone(), r["sum_cost"].copy()))
            return r

        pol._get_rollouts = spy
        # Normal.sample == torch.normal(mean, std) == randn(shape) * std + mean on the global generator;
        # the model's own eps draws (lstm.py:278) interleave with it, so record the N(0,1) draws here.
        noise = []
        orig_normal = torch.normal

        def rec_normal(mean_, std_, *a, **k):
            n = torch.randn(mean_.shape)
            noise.append(n.clone())
            return n.mul_(std_).add_(mean_)

        torch.manual_seed(123)
        torch.normal = rec_normal
        try:
            mean = pol.get_action(start, goal, 0, 0)
        finally:
            torch.normal = orig_normal
        assert len(noise) == 2
        out["ga_mean"] = mean
        for i, (a, c) in enumerate(trace):
            out[f"ga_act{i}"], out[f"ga_cost{i}"], out[f"ga_noise{i}"] = a[:, :, :2], c, noise[i]
        # self-check: the oracle's reconstruction of Normal.sample from randn
        o_mean, o_trace = orc.cem_get_action(
            sd, cfg, prob["start_img"], prob["goal_imgs"], prob["goal_masks"], T + 1, 2, N, 3, 0.03, noise,
            states=prob["states"][:, :N] if ra else None, masks=prob["masks"][:, :N] if ra else None)
        for i in range(2):
            assert np.array_equal(o_trace[i]["act_seq"], trace[i][0][:, :, :2].numpy()), "Normal.sample != mean+std*randn"
        print(tag, "oracle-vs-ref get_action max|dmean|", np.abs(o_mean - mean).max(), "gap", out["ro_gap_k"])
        save(f"cem_{tag}", **out)


def gen_dataset():
    """The numeric (non-image) part of RoboNetDataset.__getitem__ (robonet_dataset.py:173-356) on a synthetic
    trajectory: loaders, autograsp imputation, bounds, state / action preprocessing.  (The image transform is
    torchvision's, which this container lacks: no golden for the resize.)"""
    from src.dataset.robonet.robonet_dataset import RoboNetDataset, normalize, denormalize

    class FP(dict):
        attrs = {}
    g = np.random.Generator(np.random.Philox(key=[5, 5]))
    T = 9
    fp = FP(states=g.random((T, 4), dtype=np.float32), actions=g.normal(0, 0.03, (T - 1, 4)).astype(np.float32),
            qpos=g.normal(0, 1, (T, 5)).astype(np.float32),
            low_bound=np.array([0.2, -0.3, 0.05, -1.5, -1.0], np.float32),
            high_bound=np.array([0.7, 0.3, 0.35, 1.5, 1.0], np.float32))
    out = {f"in_{k}": v for k, v in fp.items()}
    # (impute_autograsp_action cannot be captured: __getitem__ hands _load_actions the SCALAR bounds low[4] / high[4]
    # and robonet_dataset.py:184 indexes them with [-1] -> IndexError in the reference itself)
    for tag, view, adim, impute in (("sawyer", "sawyer_sudri0_c0", 4, False), ("locobot", "locobot_c0", 4, False),
                                    ("franka", "franka_c0", 4, False)):
        ds = object.__new__(RoboNetDataset)
        ds._config = argparse.Namespace(robot_dim=5, robot_joint_dim=7, preprocess_action="raw")
        ds._action_dim, ds._impute_autograsp_action, ds._traj_robots = adim, impute, [view]
        low, high = ds._load_bounds(fp, view, 0)
        states = ds._load_states(fp, 2, 8)
        actions = ds._load_actions(fp, low[4], high[4], 2, 7)
        qpos = ds._load_qpos(fp, 2, 8)
        plow, phigh = ds._preprocess_bounds(low, high, 0)
        pstates = ds._preprocess_states(states, plow, phigh, view, 0)
        pact = ds._preprocess_actions(pstates, actions, plow, phigh, 0)
        out.update({f"{tag}_low": low, f"{tag}_high": high, f"{tag}_states": states, f"{tag}_actions": actions,
                    f"{tag}_qpos": qpos, f"{tag}_pstates": pstates, f"{tag}_pactions": pact.numpy()})
    x = g.normal(0, 1, (6, 5)).astype(np.float32)
    out["norm"], out["denorm"] = normalize(x, fp["low_bound"], fp["high_bound"]), denormalize(x, fp["low_bound"], fp["high_bound"])
    out["norm_in"] = x
    save("dataset_item", **out)


class _FakeSampler:
    """Stands in for the simulator-backed TrajectorySampler of the push / pick CEM variants: a cost that is a
    deterministic function of the candidate actions, so that the planner's own arithmetic is what gets pinned."""

    def __init__(self, *a, **k):
        self.calls = []

    def generate_rollouts(self, act_seq, start, goal, opt_traj=None, ret_obs=False, suppress_print=True):
        self.calls.append(act_seq.clone())
        tgt = torch.linspace(-0.3, 0.4, act_seq.shape[-1])
        cost = -((act_seq - tgt) ** 2).sum((1, 2)).double().numpy()
        return {"sum_cost": cost, "optimal_sum_cost": 0.0}

    generate_model_rollouts = generate_rollouts


def gen_sim_cem():
    """P6: the push / pick CEM variants' get_action (src/cem/push/cem.py:50-104, src/cem/pick/cem.py:50-104): initial
    belief, clamps, gripper clamp, padding, refit -- with the simulator-backed sampler replaced by _FakeSampler."""
    import src.cem.pick.cem as pick_cem
    import src.cem.push.cem as push_cem
    out = {}
    for tag, mod in (("push", push_cem), ("pick", pick_cem)):
        mod.TrajectorySampler = _FakeSampler
        cfg = argparse.Namespace(sparse_cost=False, debug_cem=False, log_dir="/tmp/x")
        pol = mod.CEMPolicy(cfg, physics="learned", horizon=4, opt_iter=3, action_candidates=40, topk=5, init_std=0.5)
        torch.manual_seed(11)
        mean = pol.get_action(State(), DemoGoalState(), 0, 0)
        out[f"{tag}_mean"] = mean
        for i, a in enumerate(pol.traj_sampler.calls):
            out[f"{tag}_act{i}"] = a.numpy()
    save("sim_cem", **out)


def gen_host_costs():
    """P4 host-side paths (losses.py:172-335): the numpy `_call` variants the MBRL loops use on environment
    observations, `img_cost_threshold`, `img_cost_world_norm`, `return_info`.  (The reference still says `np.float`,
    removed in numpy >= 1.24: aliased to the builtin for this run.)"""
    if not hasattr(np, "float"):
        np.float = float
    g = np.random.Generator(np.random.Philox(key=[8, 1]))
    a, b = g.integers(0, 256, (16, 16, 3)).astype(np.uint8), g.integers(0, 256, (16, 16, 3)).astype(np.uint8)
    m1, m2 = g.random((16, 16)) < 0.2, g.random((16, 16)) < 0.2
    s1, s2 = g.random(5), g.random(5)
    out = dict(a=a, b=b, m1=m1, m2=m2, s1=s1, s2=s2)
    for tag, rt, thr, wn in (("l2", "dense", None, True), ("l2_thr", "dense", 30, True), ("dc", "dontcare", None, True),
                             ("dc_nonorm", "dontcare", None, False), ("dc_thr", "dontcare", 30, True)):
        cf = argparse.Namespace(robot_cost_weight=0.5, world_cost_weight=2.0, reward_type=rt, img_cost_threshold=thr,
                                img_cost_world_norm=wn)
        cost = ref_losses.RobotWorldCost(cf)
        tot, info = cost(State(img=a, state=s1, mask=m1), State(img=b, state=s2, mask=m2), return_info=True)
        out[f"{tag}_total"], out[f"{tag}_robot"] = tot, info["robot_l2"]
        out[f"{tag}_world"] = info["img_dontcare" if rt == "dontcare" else "img_l2"]
    save("host_costs", **out)


def gen_robot_states():
    """8f-3: the state propagation of the analytical robot models' predict_batch
    (src/dataset/wx250s/wx250s_model.py:57-163; locobot_model.py is the same with no frame offset), run through the
    REAL reference code with the robot SDK's IK and the MuJoCo mask render replaced by inert stand-ins (neither
    influences the states)."""
    from src.dataset.wx250s.wx250s_model import WX250sAnalyticalModel
    from src.dataset.locobot.locobot_model import LocobotAnalyticalModel, PUSH_HEIGHT as LOCO_PUSH_HEIGHT
    g = np.random.Generator(np.random.Philox(key=[12, 3]))
    T, N, H, W = 6, 7, 48, 64
    low = torch.tensor([[0.015, -0.3, 0.1, 0, 0]], dtype=torch.float32)
    high = torch.tensor([[0.55, 0.3, 0.4, 1, 1]], dtype=torch.float32)
    start = torch.from_numpy(g.random(5, dtype=np.float32))
    actions = torch.from_numpy(np.clip(g.standard_normal((T, N, 5), dtype=np.float32) * 0.03, -0.05, 0.05))
    out = dict(start=start, actions=actions, low=low[0], high=high[0])

    class Env:
        def generate_masks(self, qpos):
            return [np.zeros((H, W), np.uint8) for _ in qpos]

    class Arm:
        def set_ee_pose_components(self, **k):
            return k["custom_guess"], True

    class Bot:
        arm = Arm()

    class IK:
        def ik(self, eef, alpha=None, cur_arm_config=None):
            return cur_arm_config
    for tag, Model, qd in (("wx250s", WX250sAnalyticalModel, 6), ("locobot", LocobotAnalyticalModel, 5)):
        m = object.__new__(Model)
        m._config = argparse.Namespace(device=torch.device("cpu"), image_width=W, image_height=H, preprocess_action="raw",
                                       model_use_heatmap=False)
        m.env = m.env_thick = Env()
        m._img_transform = lambda i: torch.zeros(1, H, W)
        m.bot, m.ik_solver = Bot(), IK()
        m.push_height, m.default_pitch, m.default_roll = 0.115, 1.3, 0.0
        states = torch.zeros((T + 1, N, 5))
        states[0] = start
        data = {"states": states, "qpos": torch.zeros((T + 1, N, qd)), "actions": actions, "low": low.repeat(N, 1),
                "high": high.repeat(N, 1)}
        p_states, p_masks = m.predict_batch(data, thick=True)
        assert p_masks.shape == (T + 1, N, 1, H, W)
        out[f"{tag}_states"] = p_states
    out["wx250s_push_height"], out["locobot_push_height"] = 0.115, LOCO_PUSH_HEIGHT
    save("robot_states", **out)


def gen_train_video():
    """T2: PredictionTrainer._train_video window slicing (trainer.py:259-324), sequential and `--random_snippet`
    (starts drawn from `_video_sample_rng = RandomState(seed)`, trainer.py:89), with `_train_step` recording what it
    is handed.  Frame t of the video carries the value t, so a window is identified by its content."""
    from src.prediction.trainer import PredictionTrainer
    out = {}
    for tag, snippet in (("seq", False), ("rand", True)):
        tr = object.__new__(PredictionTrainer)
        tr._config = argparse.Namespace(n_past=2, n_future=3, random_snippet=snippet, model_use_heatmap=False,
                                        load_movement_info=False, experiment="train_robonet", model_use_mask=True,
                                        model_use_robot_state=True)
        tr._video_sample_rng = np.random.RandomState(4)
        seen = []

        def step(bd, seen=seen):
            seen.append([int(bd["images"][0, 0, 0, 0, 0]), len(bd["images"]), len(bd["actions"]), len(bd["masks"]),
                         int(bd["states"][0, 0, 0]), int(bd["qpos"][-1, 0, 0])])
            return {"recon_loss": float(bd["images"][0, 0, 0, 0, 0]), "kld": 1.0}
        tr._train_step = step
        T, B = 17, 2
        ar = torch.arange(T).float()
        data = {"images": ar.view(T, 1, 1, 1, 1).expand(T, B, 3, 4, 4), "states": ar.view(T, 1, 1).expand(T, B, 5),
                "actions": ar[:-1].view(T - 1, 1, 1).expand(T - 1, B, 5), "masks": ar.view(T, 1, 1, 1, 1).expand(T, B, 1, 4, 4),
                "qpos": ar.view(T, 1, 1).expand(T, B, 5), "robot": ["a", "b"], "folder": ["f", "f"]}
        for rep in range(2):  # two videos: the snippet generator keeps its state across calls
            losses = tr._train_video(data)
            out[f"{tag}_loss{rep}"] = np.array([losses["recon_loss"], losses["kld"]])
        out[f"{tag}_windows"] = np.array(seen)
        out[f"{tag}_steps"] = tr.steps_per_train_video
    save("train_video", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["forward", "shape", "losses", "train", "traint5", "cem", "groupnorm", "eval", "sweep", "dataset",
                             "simcem", "trainvideo", "hostcosts", "heatmap", "robot"]
    if "traint5" in which:
        gen_train_t5()
    if "heatmap" in which:
        gen_heatmap()
    if "simcem" in which:
        gen_sim_cem()
    if "trainvideo" in which:
        gen_train_video()
    if "robot" in which:
        gen_robot_states()
    if "hostcosts" in which:
        gen_host_costs()
    if "dataset" in which:
        gen_dataset()
    if "forward" in which:
        gen_forward()
    if "shape" in which:
        gen_shape_pin()
    if "losses" in which:
        gen_losses()
    if "train" in which:
        gen_train()
    if "cem" in which:
        gen_cem()
    if "groupnorm" in which:
        gen_groupnorm()
    if "eval" in which:
        gen_eval()
    if "sweep" in which:
        gen_sweep()
