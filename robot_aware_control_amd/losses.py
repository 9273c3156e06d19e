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


def _is_tensor(*xs):
    return any(isinstance(x, torch.Tensor) for x in xs)


class RobotL2Cost(Cost):
    """losses.py:182-206: -||curr - goal||_2 (per row for batched tensors); 0 when a state is missing."""
    name = "robot_l2"

    def __call__(self, curr: State, goal: State):
        if curr.state is None or goal.state is None:
            return 0.0
        if _is_tensor(curr.state, goal.state):
            d = (torch.as_tensor(curr.state) - torch.as_tensor(goal.state)) ** 2
            if d.dim() not in (1, 2):
                raise NotImplementedError(f"Tensor shape {d.shape} not supported")
            return -(d.sum(1) if d.dim() == 2 else d.sum()).sqrt().cpu().numpy()
        return -np.linalg.norm(np.asarray(curr.state) - np.asarray(goal.state))


class ImgL2Cost(Cost):
    """losses.py:209-240: -||255 (curr - goal)||_2 per candidate on the GPU; numpy images (the callers in src/mbrl
    that score environment observations): -||curr - goal||_2 or, with `img_cost_threshold`, -#(|diff| > threshold)."""
    name = "img_l2"

    def _call(self, curr_img, goal_img):
        if curr_img is None or goal_img is None:
            return 0
        curr_img, goal_img = curr_img.astype(np.float64), goal_img.astype(np.float64)
        threshold = self._config.img_cost_threshold
        if threshold is None:
            return -np.linalg.norm(curr_img - goal_img)
        return -np.sum(np.abs(curr_img - goal_img) > threshold)

    def __call__(self, curr: State, goal: State):
        if _is_tensor(curr.img, goal.img):
            if curr.img is None or goal.img is None:
                return 0
            return _img_cost(0, curr.img, goal.img)
        return self._call(curr.img, goal.img)


class ImgDontcareCost(Cost):
    """losses.py:242-287: robot (curr|goal mask) pixels dropped; tensors: / #world pixels; numpy: / #world pixels only
    with `img_cost_world_norm`, optional `img_cost_threshold` counting."""
    name = "img_dontcare"

    def _call(self, curr_img, goal_img, curr_mask, goal_mask):
        if curr_img is None or goal_img is None:
            return 0
        curr_img, goal_img = curr_img.astype(np.float64), goal_img.astype(np.float64)
        total_mask = curr_mask | goal_mask
        a, b = curr_img[~total_mask], goal_img[~total_mask]
        threshold = self._config.img_cost_threshold
        loss = np.linalg.norm(a - b) if threshold is None else np.sum(np.abs(a - b) > threshold)
        if self._config.img_cost_world_norm:
            loss = loss / np.sum(~total_mask)
        return -loss

    def __call__(self, curr: State, goal: State):
        if _is_tensor(curr.img, goal.img):
            if curr.img is None or goal.img is None:
                return 0
            return _img_cost(1, curr.img, goal.img, curr.mask, goal.mask)
        return self._call(curr.img, goal.img, curr.mask, goal.mask)


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
        print_str, info = "", {}
        for w, c in ((self.robot_cost_weight, self.robot_cost), (self.world_cost_weight, self.world_cost)):
            if w == 0:
                continue
            cost = w * c(curr, goal)
            scalar = type(cost) in (np.float64, float)
            if return_info:
                if not scalar:
                    raise NotImplementedError()  # batched version: as the reference
                info[c.name] = cost
            if print_cost:
                print_str += (f"{c.name}: {cost:.4f} ," if scalar else
                              "".join(f" {c.name}: {v:.4f} ," for v in cost))
            total = total + cost
        if print_cost:
            print(print_str)
        return (total, info) if return_info else total
