"""Losses and CEM costs with the reference's call signatures (src/prediction/losses.py),
computed by the HIP kernels of librac_hip.so.  Each loss is differentiable w.r.t. `prediction`."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops
from .state import State


def _planes(t):
    return t.to(torch.float32).contiguous()


def _recon(kind, prediction, target, mask=None, robot_weight=0.0, batch_weight=None):
    out = ops.ReconLoss.apply(_planes(prediction), _planes(target), None if mask is None else _planes(mask),
                              None if batch_weight is None else _planes(batch_weight), ops.LOSS_KINDS[kind],
                              float(robot_weight))
    return out


def mse_criterion(prediction, target):
    """nn.MSELoss (losses.py:11)."""
    return _recon("mse", prediction, target)[0]


def l1_criterion(prediction, target, batch_weight=None):
    """losses.py:13-19."""
    return _recon("l1", prediction, target, None, 0.0, batch_weight)[0]


def dontcare_mse_criterion(prediction, target, mask, robot_weight):
    """losses.py:21-33."""
    return _recon("dontcare_mse", prediction, target, mask, robot_weight)[0]


def dontcare_l1_criterion(prediction, target, mask, robot_weight, batch_weight=None):
    """losses.py:35-50."""
    return _recon("dontcare_l1", prediction, target, mask, robot_weight, batch_weight)[0]


def robot_mse_criterion(prediction, target, mask):
    """losses.py:52-64."""
    with torch.no_grad():
        return _recon("mse", prediction, target, mask)[1]


def world_mse_criterion(prediction, target, mask):
    """losses.py:66-78."""
    with torch.no_grad():
        return _recon("mse", prediction, target, mask)[2]


def kl_criterion(mu1, logvar1, mu2, logvar2, bs):
    """losses.py:97-106 (inputs may be logical NCHW views of NHWC maps: the sum is layout-free)."""
    assert mu1.shape[0] == bs, f"{mu1.shape[0]} != {bs}"
    ts = (mu1, logvar1, mu2, logvar2)
    same = all(t.stride() == mu1.stride() for t in ts)
    dense = mu1.is_contiguous() or mu1.permute(0, 2, 3, 1).is_contiguous()
    if not (same and dense):  # the kernel walks memory linearly: all four must share one dense layout
        ts = tuple(t.contiguous() for t in ts)
    return ops.KLLoss.apply(*ts, bs)[0]


class Cost:
    """Generic cost interface (losses.py:172-180)."""
    name = "generic_cost"

    def __init__(self, config):
        self._config = config

    def __call__(self, curr: State, goal: State):
        raise NotImplementedError()


def _img_cost(kind, curr_img, goal_img, curr_mask=None, goal_mask=None):
    """Per-candidate image cost through rac_cem_step_tail with identity compositing."""
    n, _, H, W = curr_img.shape
    dev = curr_img.device
    x4 = torch.zeros((n, H, W, 4), device=dev)
    nxt = torch.empty((n, 3, H, W), device=dev)
    acc = torch.zeros((n,), device=dev, dtype=torch.float64)
    curr_img, goal_img = _planes(curr_img), _planes(goal_img)
    cm = None if curr_mask is None else _planes(curr_mask)
    gm = None if goal_mask is None else goal_mask.to(torch.uint8).contiguous()
    _lib.call("rac_cem_step_tail", x4.data_ptr(), curr_img.data_ptr(), None, goal_img.data_ptr(), _lib.ptr(cm),
              _lib.ptr(gm), kind, 1.0, 1, nxt.data_ptr(), acc.data_ptr(), n, H * W, _lib.stream_ptr())
    return acc.cpu().numpy().astype(np.float32)


class RobotL2Cost(Cost):
    """losses.py:182-206; the model-rollout States carry no robot state, so this is 0 there."""
    name = "robot_l2"

    def __call__(self, curr: State, goal: State):
        if curr.state is None or goal.state is None:
            return 0.0
        a, b = np.asarray(torch.as_tensor(curr.state).cpu()), np.asarray(torch.as_tensor(goal.state).cpu())
        d = (a - b) ** 2
        return -np.sqrt(d.sum(-1))


class ImgL2Cost(Cost):
    """losses.py:209-240: -||255 (curr - goal)||_2 per candidate."""
    name = "img_l2"

    def __call__(self, curr: State, goal: State):
        if curr.img is None or goal.img is None:
            return 0
        return _img_cost(0, curr.img, goal.img)


class ImgDontcareCost(Cost):
    """losses.py:242-287: robot (curr|goal mask) pixels dropped, divided by #world pixels."""
    name = "img_dontcare"

    def __call__(self, curr: State, goal: State):
        if curr.img is None or goal.img is None:
            return 0
        return _img_cost(1, curr.img, goal.img, curr.mask, goal.mask)


class RobotWorldCost(Cost):
    """Weighted robot + world cost (losses.py:290-335); zero-weight terms are skipped."""

    def __init__(self, config):
        self._config = config
        self.robot_cost_weight = config.robot_cost_weight
        self.robot_cost = RobotL2Cost(config)
        self.world_cost_weight = config.world_cost_weight
        self.world_cost = ImgDontcareCost(config) if config.reward_type == "dontcare" else ImgL2Cost(config)

    def __call__(self, curr: State, goal: State, print_cost=False, return_info=False):
        total = 0
        for w, c in ((self.robot_cost_weight, self.robot_cost), (self.world_cost_weight, self.world_cost)):
            if w == 0:
                continue
            total = total + w * c(curr, goal)
        return total
