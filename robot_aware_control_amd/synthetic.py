"""Deterministic synthetic inputs for the SVG train step and the CEM rollouts.

Shapes and distributions follow BASELINE.md section 4 / SURVEY.md section 8d:
images ~U[0,1) (T,B,3,H,W); masks Bernoulli(0.2) per pixel as float {0,1}
(T,B,1,H,W); states ~U[0,1) (T,B,R); actions ~U(-0.05,0.05) (T-1,B,A).
Counter-based (numpy Philox) so every host produces the same bytes.
"""
from __future__ import annotations

import numpy as np
import torch


def _rng(seed: int, stream: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed, stream]))


def synth_video(seed: int, T: int, B: int, H: int = 64, W: int = 64, R: int = 5, A: int = 5,
                mask_p: float = 0.2) -> dict:
    """Time-first batch in the layout `process_batch` produces
    (reference src/dataset/robonet/robonet_dataset.py:434-451)."""
    images = _rng(seed, 1).random((T, B, 3, H, W), dtype=np.float32)
    masks = (_rng(seed, 2).random((T, B, 1, H, W), dtype=np.float32) < mask_p).astype(np.float32)
    states = _rng(seed, 3).random((T, B, R), dtype=np.float32)
    actions = (_rng(seed, 4).random((T - 1, B, A), dtype=np.float32) - 0.5) * np.float32(0.1)
    return {
        "images": torch.from_numpy(images),
        "masks": torch.from_numpy(masks),
        "states": torch.from_numpy(states),
        "actions": torch.from_numpy(actions),
        "qpos": torch.from_numpy(states.copy()),
        "robot": ["sawyer"] * B,
        "folder": ["synthetic"] * B,
    }


def synth_eps(seed: int, steps: int, B: int, z: int, h: int, w: int):
    """Per time-step (eps_prior, eps_posterior) N(0,1) draws, in the order the
    reference consumes them (prior first: dynamics.py:601-608, then posterior :624)."""
    out = []
    for i in range(steps):
        g = _rng(seed, 100 + i)
        out.append((torch.from_numpy(g.standard_normal((B, z, h, w), dtype=np.float32)),
                    torch.from_numpy(g.standard_normal((B, z, h, w), dtype=np.float32))))
    return out


def synth_cem_problem(seed: int, N: int, T: int, H: int = 64, W: int = 64, with_robot: bool = False,
                      R: int = 5, mask_p: float = 0.2, goal_blend: float = 1.0) -> dict:
    """start/goal uint8 images (H,W,3), goal mask, clamped candidate actions (N,T,5);
    optional (states (T+1,N,R), masks (T+1,N,1,H,W)) standing in for the
    analytical robot model (trajectory_sampler.py:86-109)."""
    start = _rng(seed, 10).integers(0, 256, (H, W, 3), dtype=np.uint8)
    goal = _rng(seed + 1, 10).integers(0, 256, (H, W, 3), dtype=np.uint8)
    if goal_blend != 1.0:  # goal near the start frame: candidate costs then differ by more than rounding noise
        goal = ((1 - goal_blend) * start.astype(np.float32) + goal_blend * goal.astype(np.float32)).astype(np.uint8)
    act = np.zeros((N, T, 5), np.float32)
    act[:, :, :2] = np.clip(_rng(seed, 11).standard_normal((N, T, 2), dtype=np.float32) * np.float32(0.03),
                            -0.05, 0.05)
    out = {"start_img": start, "goal_imgs": [goal], "goal_masks": [np.zeros((1, H, W), bool)],
           "actions": torch.from_numpy(act)}
    if with_robot:
        out["states"] = torch.from_numpy(_rng(seed, 12).random((T + 1, N, R), dtype=np.float32))
        out["masks"] = torch.from_numpy(
            (_rng(seed, 13).random((T + 1, N, 1, H, W), dtype=np.float32) < mask_p).astype(np.float32))
        out["goal_masks"] = [(_rng(seed, 14).random((1, H, W), dtype=np.float32) < mask_p)]
    return out
