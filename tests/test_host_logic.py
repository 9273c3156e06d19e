"""CPU tests of host-side logic against vectors captured from the reference (oracle/gen_golden.py): the push / pick CEM
variants' planner arithmetic, `_train_video` window slicing (incl. `--random_snippet`), the numpy cost paths,
`reset_parameters` moments."""
import argparse
import os

import numpy as np
import pytest
import torch

from robot_aware_control_amd.state import DemoGoalState, State


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


class FakeSampler:
    """The deterministic cost of oracle/gen_golden.py:_FakeSampler."""

    def __init__(self):
        self.calls = []

    def generate_model_rollouts(self, act_seq, start, goal, opt_traj=None, ret_obs=False, suppress_print=True):
        self.calls.append(act_seq.clone())
        tgt = torch.linspace(-0.3, 0.4, act_seq.shape[-1])
        return {"sum_cost": -((act_seq - tgt) ** 2).sum((1, 2)).double().numpy(), "optimal_sum_cost": 0.0}


@pytest.mark.parametrize("tag,adim", [("push", 2), ("pick", 4)])
def test_sim_cem_variants_vs_reference(golden_dir, tag, adim):
    """src/cem/push/cem.py:50-104 and src/cem/pick/cem.py:50-104: clamp to [-1, 1], no do-nothing candidate, the pick
    variant's initial belief and gripper clamp, zero padding of the push actions to 5-D, refit -- same Normal draws
    (torch.manual_seed) -> the candidates handed to the sampler and the returned mean are bit-identical."""
    from robot_aware_control_amd.cem import SimCEMPolicy
    g = load(golden_dir, "sim_cem")
    cfg = argparse.Namespace(sparse_cost=False, debug_cem=False, log_dir="/tmp/x", device="cpu", robot_cost_weight=0.0,
                             world_cost_weight=1.0, reward_type="dense", action_dim=5 if adim == 2 else 4)
    pol = SimCEMPolicy(cfg, physics="learned", horizon=4, opt_iter=3, action_candidates=40, topk=5, init_std=0.5,
                       action_dim=adim, model=object())
    pol.traj_sampler = FakeSampler()
    torch.manual_seed(11)
    mean = pol.get_action(State(), DemoGoalState(), 0, 0)
    assert mean.shape == (3, adim) and np.array_equal(mean, g[f"{tag}_mean"])
    for i, a in enumerate(pol.traj_sampler.calls):
        assert np.array_equal(a.numpy(), g[f"{tag}_act{i}"]), i
    if adim == 4:
        assert float(pol.traj_sampler.calls[0][:, :, -1].max()) <= 0 and float(pol.traj_sampler.calls[0][:, :, -1].min()) >= -0.01
    with pytest.raises(NotImplementedError):
        SimCEMPolicy(cfg, physics="gt", model=object())
    with pytest.raises(NotImplementedError):  # the reference's default (push/cem.py:24) is "gt": never silently "learned"
        SimCEMPolicy(cfg, model=object())


@pytest.mark.parametrize("tag,snippet", [("seq", False), ("rand", True)])
def test_train_video_windows_vs_reference(golden_dir, tag, snippet):
    """PredictionTrainer._train_video (trainer.py:259-324): floor(T / window) windows, sequential or drawn from
    `_video_sample_rng` (RandomState(seed), state kept across videos); losses averaged over the windows."""
    from robot_aware_control_amd.trainer import PredictionTrainer
    g = load(golden_dir, "train_video")
    tr = object.__new__(PredictionTrainer)
    tr._config = argparse.Namespace(n_past=2, n_future=3, random_snippet=snippet, model_use_heatmap=False,
                                    load_movement_info=False, experiment="train_robonet", model_use_mask=True,
                                    model_use_robot_state=True)
    tr._video_sample_rng = np.random.RandomState(4)
    seen = []

    def step(bd):
        seen.append([int(bd["images"][0, 0, 0, 0, 0]), len(bd["images"]), len(bd["actions"]), len(bd["masks"]),
                     int(bd["states"][0, 0, 0]), int(bd["qpos"][-1, 0, 0])])
        return {"recon_loss": float(bd["images"][0, 0, 0, 0, 0]), "kld": 1.0}
    tr._train_step = step
    T, B = 17, 2
    ar = torch.arange(T).float()
    data = {"images": ar.view(T, 1, 1, 1, 1).expand(T, B, 3, 4, 4), "states": ar.view(T, 1, 1).expand(T, B, 5),
            "actions": ar[:-1].view(T - 1, 1, 1).expand(T - 1, B, 5), "masks": ar.view(T, 1, 1, 1, 1).expand(T, B, 1, 4, 4),
            "qpos": ar.view(T, 1, 1).expand(T, B, 5), "robot": ["a", "b"], "folder": ["f", "f"]}
    for rep in range(2):
        losses = tr._train_video(data)
        np.testing.assert_allclose([losses["recon_loss"], losses["kld"]], g[f"{tag}_loss{rep}"], rtol=1e-12)
    assert np.array_equal(np.array(seen), g[f"{tag}_windows"]) and tr.steps_per_train_video == int(g[f"{tag}_steps"])


def test_finetune_windows_use_the_robot_model():
    """finetune_* experiments (trainer.py:294-319): states and masks of every window come from `robot_model`."""
    from robot_aware_control_amd.trainer import PredictionTrainer
    tr = object.__new__(PredictionTrainer)
    tr._config = argparse.Namespace(n_past=1, n_future=2, random_snippet=False, model_use_heatmap=False,
                                    load_movement_info=False, experiment="finetune_locobot", model_use_mask=True,
                                    model_use_robot_state=True, preprocess_action="raw")
    T, B = 6, 2
    data = {"images": torch.zeros(T, B, 3, 4, 4), "states": torch.zeros(T, B, 5), "actions": torch.zeros(T - 1, B, 5),
            "masks": torch.zeros(T, B, 1, 4, 4), "qpos": torch.zeros(T, B, 5), "robot": ["a", "b"], "folder": ["f", "f"],
            "low": torch.zeros(B, 5), "high": torch.ones(B, 5)}
    tr._train_step = lambda bd: {"recon_loss": float(bd["states"].sum() + bd["masks"].sum())}
    tr.robot_model = None
    with pytest.raises(NotImplementedError):
        tr._train_video(data)

    class Robot:
        def predict_batch(self, bd):
            assert set(("low", "high", "qpos", "actions")) <= set(bd)
            return torch.ones_like(bd["states"]), torch.ones_like(bd["masks"])
    tr.robot_model = Robot()
    assert tr._train_video(data)["recon_loss"] == 3 * 2 * 5 + 3 * 2 * 16
    # evaluation (trainer.py:520-547): predicted states / masks drive the rollout, the TRUE masks score it
    tr._config.n_eval = 3
    tr._eval_step = lambda bd, autoreg: {"err": float(bd["states"].sum() + bd["pred_masks"].sum() + bd["masks"].sum())}
    assert tr._eval_video(data, autoregressive=True)["err"] == 3 * 2 * 5 + 3 * 2 * 16
    tr.robot_model = None
    with pytest.raises(NotImplementedError):
        tr._eval_video(data)


def test_host_cost_paths_vs_reference(golden_dir):
    """RobotWorldCost on numpy observations (losses.py:182-335): image L2 / dontcare, thresholds, world-norm, info."""
    from robot_aware_control_amd.losses import RobotWorldCost
    g = load(golden_dir, "host_costs")
    for tag, rt, thr, wn in (("l2", "dense", None, True), ("l2_thr", "dense", 30, True), ("dc", "dontcare", None, True),
                             ("dc_nonorm", "dontcare", None, False), ("dc_thr", "dontcare", 30, True)):
        cf = argparse.Namespace(robot_cost_weight=0.5, world_cost_weight=2.0, reward_type=rt, img_cost_threshold=thr,
                                img_cost_world_norm=wn)
        tot, info = RobotWorldCost(cf)(State(img=g["a"], state=g["s1"], mask=g["m1"]),
                                       State(img=g["b"], state=g["s2"], mask=g["m2"]), return_info=True)
        assert tot == float(g[f"{tag}_total"]) and info["robot_l2"] == float(g[f"{tag}_robot"])
        assert info["img_dontcare" if rt == "dontcare" else "img_l2"] == float(g[f"{tag}_world"])
    cf.robot_cost_weight = 0.0
    assert RobotWorldCost(cf)(State(img=g["a"], mask=g["m1"]), State(img=g["b"], mask=g["m2"])) == float(g["dc_thr_world"])


def test_reset_parameters_moments():
    """init_weights (base.py:26-36): conv W ~ N(0, 0.02), conv b = 0, BatchNorm gamma ~ N(1, 0.02), beta = 0;
    GroupNorm (NormConvLSTMCell) keeps its default ones / zeros."""
    from robot_aware_control_amd import model as M
    cf = argparse.Namespace(image_width=64, image_height=64, channels=3, g_dim=64, z_dim=16, action_dim=5, robot_dim=5,
                            batch_size=2, model_use_mask=True, model_use_future_mask=True, model_use_heatmap=False,
                            model_use_future_heatmap=False, model_use_robot_state=True, model_use_future_robot_state=False,
                            lstm_group_norm=True, last_frame_skip=True, device=torch.device("cpu"))
    torch.manual_seed(0)
    m = M.SVGConvModel(cf)
    convw, bnw = [], []
    for mod in m.modules():
        if isinstance(mod, M._Conv):
            convw.append(mod.weight.detach().flatten())
            assert mod.bias is None or float(mod.bias.abs().max()) == 0.0
        elif isinstance(mod, M._BatchNorm):
            bnw.append(mod.weight.detach())
            assert float(mod.bias.abs().max()) == 0.0 and float(mod.running_mean.abs().max()) == 0.0
            assert float((mod.running_var - 1).abs().max()) == 0.0
        elif isinstance(mod, M._GroupNorm):
            assert float((mod.weight - 1).abs().max()) == 0.0 and float(mod.bias.abs().max()) == 0.0
    w, g = torch.cat(convw), torch.cat(bnw)
    assert w.numel() > 1e6 and abs(float(w.mean())) < 1e-4 and abs(float(w.std()) - 0.02) < 2e-4
    assert abs(float(g.mean()) - 1) < 2e-3 and abs(float(g.std()) - 0.02) < 2e-3
    # parameters are views of ONE flat buffer (the fused Adam / all-reduce contract)
    flat, grad = m.flat_parameters()
    used = sum((p.numel() + 3) // 4 * 4 for p in m.parameters())  # 16-byte aligned views, the buffer padded to whole 4 KB
    assert (used + 1023) // 1024 * 1024 == flat.numel() == grad.numel() and float(flat[used:].abs().max()) == 0.0


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: the script starts `torch.distributed.run` as a CHILD process
    with the same arguments and exits with its return code -- before torch or the HIP library is imported (bench.py calls
    this ahead of those imports); with WORLD_SIZE set, or N = 1, it is a rank itself."""
    import importlib
    import subprocess
    import sys
    import types
    bench = importlib.import_module("bench")
    calls = []
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None: (calls.append((cmd, env)), types.SimpleNamespace(returncode=7))[1])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    for argv in (["bench.py", "--gpus", "4", "--steps", "3"], ["bench.py", "--steps", "3", "--gpus=4"]):
        calls.clear()
        monkeypatch.setattr(sys, "argv", argv)
        with pytest.raises(SystemExit) as e:
            bench._launch_ranks_if_needed()
        assert e.value.code == 7 and len(calls) == 1
        cmd, env = calls[0]
        assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
        assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
        assert cmd[-len(argv) + 1:] == argv[1:] and cmd[-len(argv)].endswith("bench.py")
        assert env.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    calls.clear()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    bench._launch_ranks_if_needed()  # one GPU: this process is the benchmark
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench._launch_ranks_if_needed()  # launched by torch.distributed.run: a rank
    assert not calls
