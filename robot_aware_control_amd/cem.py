"""CEMPolicy: cross-entropy-method planner over the frozen SVG model (API of reference
src/cem/cem.py:14-111): sample N action sequences, roll them out in batches on the GPU(s),
keep the top-K by summed cost, refit mean / std, repeat."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist
from torch.distributions.normal import Normal

from . import parallel_env
from .state import DemoGoalState, State
from .trajectory_sampler import TrajectorySampler


class CEMPolicy(object):
    """Given the current state and goal images, use CEM to find the best actions."""

    def __init__(self, cfg, model, horizon=5, opt_iter=10, action_candidates=100, topk=5, init_std=1.0,
                 cam_ext=None, franka_ik=None, wx250s_bot=None, push_height=None, default_pitch=None,
                 default_roll=None, robot_model=None):
        self.horizon = horizon
        self.optimization_iter = opt_iter
        self.num_actions = action_candidates
        self.K = topk
        self.init_std = init_std
        self.sparse_cost = cfg.sparse_cost
        self.action_dim = 2
        self.cfg = cfg
        self.model = model
        self.traj_sampler = TrajectorySampler(cfg, self.model, cam_ext=cam_ext, franka_ik=franka_ik,
                                              wx250s_bot=wx250s_bot, push_height=push_height,
                                              default_pitch=default_pitch, default_roll=default_roll,
                                              robot_model=robot_model)
        self.clamp = 0.05  # real-robot planar displacements (cem.py:85); the sim variants use 1.0
        self.null_candidate = True  # cem.py:82-83; the sim variants do not force a do-nothing candidate
        self.gripper_clamp = None  # pick variant: (-0.01, 0) on the last action dim (pick/cem.py:88)
        self.plot_rollouts = cfg.debug_cem
        if self.plot_rollouts:
            self.debug_cem_dir = cfg.log_dir
            os.makedirs(self.debug_cem_dir, exist_ok=True)
        self.trace = None  # set to [] to record (act_seq, sum_cost, top_idx, mean, std) per iteration

    def _sample(self, mean, std, N, noise=None):
        """`Normal(mean, std).sample((N,))` (cem.py:80-81); every rank must see the same candidates,
        so rank 0's draw is broadcast when running sharded."""
        if noise is not None:
            act_seq = mean + std * noise
        else:
            act_seq = Normal(mean, std).sample((N,))
        if parallel_env.active():
            dev = torch.device(self.cfg.device)
            buf = act_seq.to(dev) if dist.get_backend() == "nccl" else act_seq.clone()
            dist.broadcast(buf, src=0)
            act_seq = buf.cpu()
        return act_seq

    def get_action(self, start, goal, ep_num, step, opt_traj=None, noise=None):
        """Returns the refit mean action sequence, np.ndarray (horizon-1, 2).
        `noise`: optional list of N(0,1) draws (N,T-1,2) per iteration (parity tests)."""
        T, A, N = self.horizon, self.action_dim, self.num_actions
        self.ep_num, self.step = ep_num, step
        mean, std = self._init_belief(T, A)
        mean_top_costs = []
        rollouts = {}
        for i in range(self.optimization_iter):
            act_seq = self._sample(mean, std, N, None if noise is None else noise[i])
            if i == 0 and self.null_candidate:
                act_seq[-1] = 0  # always keep a "do nothing" candidate (cem.py:82-83)
            act_seq.clamp_(-self.clamp, self.clamp)
            if self.gripper_clamp is not None:
                act_seq[:, :, -1].clamp_(*self.gripper_clamp)
            pad = max(0, getattr(self.cfg, "action_dim", 5) - A)
            padded = torch.cat([act_seq, torch.zeros((N, T - 1, pad))], 2) if pad else act_seq
            last = i == self.optimization_iter - 1
            rollouts = self._get_rollouts(padded, start, goal, opt_traj if last else None,
                                          self.plot_rollouts and last)
            costs = torch.from_numpy(np.asarray(rollouts["sum_cost"]))
            top_costs, top_idx = costs.topk(self.K)
            top_act_seq = torch.index_select(act_seq, dim=0, index=top_idx)
            mean_top_costs.append(f"{top_costs.mean():.3f}")
            std, mean = torch.std_mean(top_act_seq, dim=0)
            std = torch.max(0.001 * torch.ones_like(std), std)
            if self.trace is not None:
                self.trace.append({"act_seq": act_seq.numpy().copy(), "sum_cost": costs.numpy().copy(),
                                   "top_idx": top_idx.numpy().copy(), "mean": mean.numpy().copy(),
                                   "std": std.numpy().copy()})
        self.mean_top_costs = mean_top_costs
        return mean.numpy()

    def _init_belief(self, T, A):
        """Initial action-sequence belief N(0, init_std) of shape (T-1, A) (cem.py:72-73)."""
        return torch.zeros(T - 1, A), torch.ones(T - 1, A) * self.init_std

    def _get_rollouts(self, act_seq, start: State, goal: DemoGoalState, opt_traj=None, plot=False):
        return self.traj_sampler.generate_model_rollouts(act_seq, start, goal, ret_obs=self.plot_rollouts,
                                                         opt_traj=opt_traj, suppress_print=True)


class SimCEMPolicy(CEMPolicy):
    """Constructor / clamp conventions of the simulator variants (reference src/cem/push/cem.py:15-104,
    src/cem/pick/cem.py): `CEMPolicy(cfg, physics="learned", horizon, opt_iter, action_candidates, topk, init_std)`
    loads the model itself from `cfg.dynamics_model_ckpt` and clamps actions to [-1, 1]; `action_dim` is 2 (push)
    or 4 (pick).  Only the `physics="learned"` branch runs here -- `"gt"` steps MuJoCo on the CPU and is outside
    the accelerated path.  The DEFAULT is the reference's (`physics="gt"`, push/cem.py:24): a caller that relies on it
    gets the clear NotImplementedError below, never silently the other branch.  Per-candidate masks/states come from
    `robot_model` (see TrajectorySampler)."""

    def __init__(self, cfg, physics="gt", horizon=5, opt_iter=10, action_candidates=100, topk=5, init_std=1.0,
                 action_dim=2, robot_model=None, model=None):
        if physics != "learned":
            raise NotImplementedError("physics='gt' rolls candidates through MuJoCo on the CPU (reference "
                                      "src/cem/push/trajectory_sampler.py:60-167); use the reference for that branch")
        if model is None:
            from .model import SVGConvModel
            model = SVGConvModel(cfg)
            if getattr(cfg, "dynamics_model_ckpt", None):
                ckpt = torch.load(cfg.dynamics_model_ckpt, map_location=cfg.device)
                model.load_state_dict(ckpt["model"])
            model.eval()
        super().__init__(cfg, model, horizon=horizon, opt_iter=opt_iter, action_candidates=action_candidates,
                         topk=topk, init_std=init_std, robot_model=robot_model)
        self.action_dim = action_dim
        self.physics = physics
        self.clamp = 1.0
        self.null_candidate = False
        if action_dim == 4:
            self.gripper_clamp = (-0.01, 0.0)

    def _init_belief(self, T, A):
        mean, std = super()._init_belief(T, A)
        if A == 4:  # pick variant (pick/cem.py:67-72): narrower x, gripper command in [-0.01, 0]
            std[:, 0] = 0.2
            mean[:, -1] = -0.005
            std[:, -1] = 0.005
        return mean, std
