"""Batch layout contract of the dataloader (reference src/dataset/robonet/robonet_dataset.py:434-467).

The hdf5 datasets themselves are I/O and stay the reference's; the hot path only depends on
`process_batch`: batch-first (B, T, ...) tensors become time-first (T, B, ...) tensors on the device."""
from __future__ import annotations

import torch

TRANSPOSE_KEYS = ("qpos", "images", "states", "actions", "masks", "heatmaps", "raw_actions", "raw_states")


def process_batch(data: dict, device) -> dict:
    """Changes tensor idx from batch-first to time-first and moves it to `device` (non-blocking H2D:
    pin the loader's memory to overlap the copy with the previous train step)."""
    for k in TRANSPOSE_KEYS:
        if k in data:
            data[k] = data[k].transpose_(1, 0).to(device, non_blocking=True)
    return data


def get_batch(loader, device):
    """Infinite batch generator over a dataloader."""
    while True:
        for data in loader:
            yield process_batch(data, device)


class SyntheticVideoDataset(torch.utils.data.Dataset):
    """Batch-first synthetic videos in the dataset's item layout (images (T,3,H,W), masks (T,1,H,W), ...),
    for `--data_root synthetic` runs and dataloader plumbing tests."""

    def __init__(self, n: int, T: int, H: int = 64, W: int = 64, R: int = 5, A: int = 5, seed: int = 0):
        self.n, self.T, self.H, self.W, self.R, self.A, self.seed = n, T, H, W, R, A, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        from .synthetic import synth_video
        v = synth_video(self.seed * 100003 + i, self.T, 1, self.H, self.W, self.R, self.A)
        out = {k: v[k][:, 0] for k in ("images", "masks", "states", "actions", "qpos")}
        out["robot"] = "sawyer"
        out["folder"] = "synthetic"
        return out
