"""Write a tiny RoboNet-shaped dataset of synthetic trajectories (the fixture of tests/test_data_path.py; also handy
for a first end-to-end run of `python -m src.prediction.multirobot_trainer --data_root <dir> ...` without the 36 GB
of RoboNet).  Layout = what robonet_dataloaders.py:21-199 scans:

    <root>/sawyer_views/sudri0_c0/traj_000.npz ...      <root>/widowx_views/widowx1_c0/...    <root>/baxter_views/left_c0/...

Every file holds the datasets of a RoboNet hdf5 under the same names (frames uint8 (T,64,85,3), mask (T,64,85),
states (T,5) normalised, actions (T-1,4), qpos (T,7), low_bound / high_bound (5,)) plus `attr_robot`: `.npz` instead of
HDF5 because h5py is not part of every image; robot_aware_control_amd.data reads both.

    python tools/make_synthetic_robonet.py /tmp/robonet_synth --per-view 6 --length 12
"""
import argparse
import os

import numpy as np

VIEWS = {"sawyer": ("sawyer_views", ["sudri0_c0", "sudri2_c1"]), "widowx": ("widowx_views", ["widowx1_c0"]),
         "baxter": ("baxter_views", ["left_c0"])}


LOCOBOT = ("locobot_views", ["c0", "c1"])  # the unseen robot of the zero-shot transfer evaluation (metric states, no bounds)


def write(root, per_view=4, length=12, seed=0, h=64, w=85, locobot=0):
    """`locobot` > 0: also that many trajectories per locobot view (locobot_singleview_dataloader.py's layout: metric
    5-dim states, 5 joint angles, no low_bound / high_bound, no robot attribute); not counted in the return value."""
    for j, view in enumerate(LOCOBOT[1] if locobot else []):
        d = os.path.join(root, LOCOBOT[0], view)
        os.makedirs(d, exist_ok=True)
        for i in range(locobot):
            g = np.random.Generator(np.random.Philox(key=[seed, 10_000 + 100 * j + i]))
            T = length + int(g.integers(0, 3))
            xy = np.cumsum(g.normal(0, 0.01, (T, 2)), 0) + np.array([0.3, 0.0])
            states = np.concatenate([xy, np.full((T, 1), 0.12), np.zeros((T, 2))], 1).astype(np.float32)
            mask = np.zeros((T, h, w), np.uint8)
            for t in range(T):
                cy, cx = int(30 + 40 * xy[t, 1]), int(20 + 100 * (xy[t, 0] - 0.2))
                mask[t, max(0, cy - 6):cy + 6, max(0, cx - 9):cx + 9] = 1
            np.savez(os.path.join(d, f"traj_{i:03d}.npz"), observations=g.integers(0, 256, (T, h, w, 3), dtype=np.uint8),
                     masks=mask, states=states, actions=np.diff(states[:, :4], axis=0).astype(np.float32),
                     qpos=g.normal(0, 1, (T, 5)).astype(np.float32))
    n = 0
    for robot, (sub, views) in VIEWS.items():
        for view in views:
            d = os.path.join(root, sub, view)
            os.makedirs(d, exist_ok=True)
            for i in range(per_view):
                g = np.random.Generator(np.random.Philox(key=[seed, n]))
                T = length + int(g.integers(0, 3))
                low = np.array([0.2, -0.3, 0.05, -1.5, -1.0], np.float32) + g.normal(0, 0.01, 5).astype(np.float32)
                high = low + np.array([0.5, 0.6, 0.3, 3.0, 2.0], np.float32)
                frames = g.integers(0, 256, (T, h, w, 3), dtype=np.uint8)
                mask = np.zeros((T, h, w), np.uint8)
                for t in range(T):  # a moving blob: the "robot"
                    cy, cx = int(10 + 2 * t + g.integers(0, 3)), int(20 + 3 * t)
                    mask[t, max(0, cy - 6):cy + 6, max(0, cx - 9):cx + 9] = 1
                np.savez(os.path.join(d, f"traj_{i:03d}.npz"), frames=frames, mask=mask,
                         states=g.random((T, 5), dtype=np.float32), actions=g.normal(0, 0.03, (T - 1, 4)).astype(np.float32),
                         qpos=g.normal(0, 1, (T, 7)).astype(np.float32), low_bound=low, high_bound=high,
                         attr_robot=np.array(robot))
                n += 1
    return n


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--per-view", type=int, default=4)
    ap.add_argument("--length", type=int, default=12)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    print(write(a.root, a.per_view, a.length, a.seed), "trajectories written under", a.root)
