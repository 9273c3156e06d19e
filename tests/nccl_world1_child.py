"""Child process of tests/test_gpu_nccl.py: ONE rank on the box's one MI355X, backend "nccl" (= RCCL), with
RAC_DIST_FORCE=1 so that every collective call site of the path executes even at world size 1 (two ranks cannot share a
device under RCCL): the planner's candidate broadcast (cem.py `_sample`), the cost all-gather on device tensors
(trajectory_sampler.py `gather_costs`), the trainer's parameter broadcast, `GradReducer`'s async slices + `finish`, and
the sharded optimiser's reduce-scatter / all-gather.  Results are compared with the same calls with the collectives off.
Prints one JSON line."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", RAC_DIST_FORCE="1")
    os.environ.setdefault("MASTER_PORT", "29541")
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    calls = []  # (collective, device type of its tensor)

    def spy(name):
        real = getattr(dist, name)

        def wrapped(*a, **k):
            t = a[0][0] if isinstance(a[0], (list, tuple)) else a[0]
            calls.append((name, t.device.type))
            return real(*a, **k)
        setattr(dist, name, wrapped)
    for name in ("broadcast", "all_gather", "all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor"):
        spy(name)

    from oracle import svg_oracle as orc
    from robot_aware_control_amd import parallel_env, synthetic as syn
    from robot_aware_control_amd import trainer as trainer_mod
    from robot_aware_control_amd.cem import CEMPolicy
    from robot_aware_control_amd.model import SVGConvModel
    from robot_aware_control_amd.state import DemoGoalState, State
    from robot_aware_control_amd.trainer import PredictionTrainer
    assert parallel_env.active() and dist.get_backend() == "nccl"
    out = {"backend": dist.get_backend()}
    flags = dict(model_use_mask=False, model_use_future_mask=False, model_use_robot_state=False, reconstruction_loss="l1")
    cfg = orc.Cfg(g_dim=64, z_dim=16, batch_size=2, n_past=1, n_future=2, lr=1e-4, candidates_batch_size=4,
                  sample_mean=True, reward_type="dense", topk=3, **flags)
    d = dict(cfg.__dict__)
    d.update(device=dev, debug_cem=False, log_dir="/tmp/rac_nccl", img_cost_threshold=None, img_cost_world_norm=True,
             experiment="train_robonet", robot_joint_dim=5, load_movement_info=False, movement_weight=1.0,
             scheduled_sampling=False, scheduled_sampling_k=4000, model="svg", optimizer="adam", seed=0, wandb=False,
             cem_shard=True, ddp_bucket_mb=1, dynamics_model_ckpt=None, ddp_shard_optimizer=False)
    ns = argparse.Namespace(**d)

    # ---- planner: candidate broadcast + cost all-gather through RCCL ----
    model = SVGConvModel(ns)
    model.load_state_dict({k: v.clone() for k, v in orc.make_weights(cfg, seed=9, action_gain=200.0).items()})
    model.eval()
    N, T = 11, 4
    prob = syn.synth_cem_problem(seed=5, N=N, T=T, goal_blend=0.15)
    start, goal = State(img=prob["start_img"]), DemoGoalState(imgs=prob["goal_imgs"], masks=prob["goal_masks"])
    pol = CEMPolicy(ns, model, horizon=T + 1, opt_iter=2, action_candidates=N, topk=3, init_std=0.03)
    n0 = len(calls)
    sh = pol.traj_sampler.generate_model_rollouts(prob["actions"].clone(), start, goal)["sum_cost"]
    out["gather_on_device"] = ("all_gather", "cuda") in calls[n0:]
    ns.cem_shard = False
    one = pol.traj_sampler.generate_model_rollouts(prob["actions"].clone(), start, goal)["sum_cost"]
    ns.cem_shard = True
    out["cem_equal"] = bool(np.array_equal(sh, one))
    n0 = len(calls)
    torch.manual_seed(3)
    act = pol.get_action(start, goal, 0, 0)
    out["broadcast_on_device"] = calls[n0:].count(("broadcast", "cuda")) == 2  # one per CEM iteration
    os.environ["RAC_DIST_FORCE"] = "0"
    torch.manual_seed(3)
    act1 = pol.get_action(start, goal, 0, 0)
    os.environ["RAC_DIST_FORCE"] = "1"
    out["action_equal"] = bool(np.array_equal(act, act1))

    # ---- trainer: parameter broadcast, GradReducer (async slices as the ConvLSTM wgrads finish + finish()) ----
    def fresh(shard_opt):
        ns.ddp_shard_optimizer = shard_opt
        tr = PredictionTrainer(ns)
        tr.model.load_state_dict({k: v.clone() for k, v in orc.make_weights(cfg, seed=1, randomize_bn_stats=False).items()})
        tr.model.train()
        tr.model.eps_source = lambda shape: torch.zeros(shape)
        return tr
    n0 = len(calls)
    tr = fresh(False)
    out["param_broadcast"] = ("broadcast", "cuda") in calls[n0:]
    data = syn.synth_video(seed=30, T=3, B=2)
    n0 = len(calls)
    l_dist = tr._train_step(data)
    out["allreduce_slices"] = calls[n0:].count(("all_reduce", "cuda"))
    g_dist = tr.model.flat_parameters()[1].clone()
    p_dist = tr.model.flat_parameters()[0].clone()
    os.environ["RAC_DIST_FORCE"] = "0"
    tr1 = fresh(False)
    l_one = tr1._train_step(data)
    os.environ["RAC_DIST_FORCE"] = "1"
    g_one, p_one = tr1.model.flat_parameters()[1], tr1.model.flat_parameters()[0]
    out["grad_rel_diff"] = float((g_dist - g_one).norm() / g_one.norm())
    out["param_rel_diff"] = float((p_dist - p_one).norm() / p_one.norm())
    out["loss_equal"] = all(abs(l_dist[k] - l_one[k]) <= 1e-6 * abs(l_one[k]) for k in l_one)

    # ---- sharded optimiser: reduce-scatter of the gradient, Adam on this rank's slices, all-gather of the parameters ----
    n0 = len(calls)
    tr2 = fresh(True)
    waits = []
    real_wait = tr2.optimizer.wait_params
    tr2.optimizer.wait_params = lambda upto=None: (waits.append(upto), real_wait(upto))[1]
    tr2._train_step(data)
    from robot_aware_control_amd import ops as _ops
    out["gate_armed"] = _ops.PARAM_GATE is tr2.optimizer and len(tr2.optimizer._pending) > 0
    tr2._train_step(data)  # the second step waits for the first one's parameter all-gather, in two stages:
    # the encoder's buckets before its first kernel, everything else behind the encoder's forward pass
    enc_end = tr2.model._encoder_extent()
    out["staged_wait"] = enc_end in waits and waits.index(enc_end) < len(waits) - 1 and None in waits[waits.index(enc_end):]
    tr2.optimizer.wait_params()
    out["gate_released"] = _ops.PARAM_GATE is None
    seen = calls[n0:]
    out["reduce_scatter"] = seen.count(("reduce_scatter_tensor", "cuda"))
    out["param_allgather"] = seen.count(("all_gather_into_tensor", "cuda"))
    os.environ["RAC_DIST_FORCE"] = "0"
    fused_log = []
    real_fused = _ops.fused_adam_step
    _ops.fused_adam_step = lambda *a, **k: (fused_log.append(real_fused(*a, **k)), fused_log[-1])[1]
    tr3 = fresh(False)
    tr3._train_step(data)
    tr3._train_step(data)
    os.environ["RAC_DIST_FORCE"] = "1"
    a, b = tr2.model.flat_parameters()[0], tr3.model.flat_parameters()[0]
    out["sharded_param_rel_diff"] = float((a - b).norm() / b.norm())
    # the yardstick: the SAME plain run twice (the fp64 atomics of the BatchNorm statistics and the fp32 atomics of the
    # bias gradients are order dependent; Adam turns that noise into +-lr steps where a gradient is only noise)
    os.environ["RAC_DIST_FORCE"] = "0"
    tr4 = fresh(False)
    tr4._train_step(data)
    tr4._train_step(data)
    os.environ["RAC_DIST_FORCE"] = "1"
    c = tr4.model.flat_parameters()[0]
    out["plain_rerun_rel_diff"] = float((c - b).norm() / b.norm())
    out["fused_adam_taken"] = fused_log  # [run 1 step 1, step 2, run 2 step 1, step 2]
    _ops.fused_adam_step = real_fused
    upd = (b - p_one).norm() / p_one.norm()  # (size of one-to-two optimiser steps, for scale)
    out["two_step_update_rel"] = float(upd)
    torch.cuda.synchronize()
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
