"""The data path that feeds the train step: trajectory files -> (B, T, ...) batches -> time-first tensors on the device.

Drop-in for the reference's `src/dataset/robonet/robonet_dataset.py` (`RoboNetDataset`, `process_batch`, `get_batch`,
`normalize`, `denormalize`) and `robonet_dataloaders.py` (`create_loaders`): same constructor arguments, same item
dictionary (keys, shapes, dtypes), same window sampling, state / action / bound preprocessing and mask binarisation,
same file discovery, shuffling and train / test split.

What is different, for a trainer that consumes ~2 800 frames/s per GPU:
  * frames are resized with one batched `interpolate` call per trajectory instead of one PIL / torchvision call per
    frame (bilinear, align_corners=False, no antialias: what `tf.Resize` does to a tensor in the torchvision the
    reference pins);
  * `get_batch` runs a DEVICE PREFETCHER: the next batch's pinned host tensors are copied on a side HIP stream and
    transposed to time-first there while the current step computes (the reference transposes on the host and copies on
    the compute stream);
  * trajectory files may be HDF5 (`h5py`, imported when the first `.hdf5` file is opened) or `.npz` archives with the
    same dataset names (the format of `tools/make_synthetic_robonet.py`, used by the tests: this container has no h5py).
"""
from __future__ import annotations

import os
import pickle
import random
import threading
from queue import Queue

import numpy as np
import torch
import torch.nn.functional as F
import torch.utils.data as data

TRANSPOSE_KEYS = ("qpos", "images", "states", "actions", "masks", "heatmaps", "raw_actions", "raw_states")

# robonet_dataloaders.py:13-18
BAXTER_TRAIN_DIRS = ["left_c0"]
WIDOWX_TRAIN_DIRS = ["widowx1_c0"]
SAWYER_TRAIN_DIRS = ["sudri0_c0", "sudri0_c1", "sudri0_c2", "sudri2_c0", "sudri2_c1", "sudri2_c2", "vestri_table2_c0",
                     "vestri_table2_c1", "vestri_table2_c2"]
TRAJ_EXTENSIONS = (".hdf5", ".npz")


def denormalize(states, low, high):
    """robonet_dataset.py:470-473."""
    states = states * (high - low)
    states = states + low
    return states


def normalize(states, low, high):
    """robonet_dataset.py:476-479."""
    states = states - low
    states = states / (high - low)
    return states


# --------------------------------------------------------------------------- #
# trajectory files
# --------------------------------------------------------------------------- #
class _NpzTrajectory:
    """An `.npz` archive behind the part of the h5py.File interface the dataset uses (`in`, `[name]`, `.attrs`)."""

    def __init__(self, path):
        self._z = np.load(path, allow_pickle=False)
        self.attrs = {k[5:]: str(self._z[k]) for k in self._z.files if k.startswith("attr_")}

    def __contains__(self, k):
        return k in self._z.files

    def __getitem__(self, k):
        return self._z[k]

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self._z.close()


def open_trajectory(path):
    if path.endswith(".npz"):
        return _NpzTrajectory(path)
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - depends on the deployment image
        raise ImportError(f"{path}: reading HDF5 trajectories needs h5py (or convert them to .npz with the same "
                          f"dataset names)") from e
    return h5py.File(path, "r")


def _camera_dicts():
    """Camera calibration tables live in the reference tree (src/utils/camera_calibration.py: measured extrinsics of
    the authors' rigs); only the `camera_*` action preprocessing needs them."""
    from src.utils.camera_calibration import camera_to_world_dict, world_to_camera_dict
    return world_to_camera_dict, camera_to_world_dict


# --------------------------------------------------------------------------- #
# image preprocessing (robonet_dataset.py:257-300)
# --------------------------------------------------------------------------- #
def _to_tensor(x: np.ndarray) -> torch.Tensor:
    """tf.ToTensor on a stack: uint8 (T,H,W,C) -> float (T,C,H,W) / 255; float (T,H,W) -> (T,1,H,W) unscaled."""
    t = torch.from_numpy(np.ascontiguousarray(x))
    if t.dim() == 3:
        t = t.unsqueeze(-1)
    t = t.permute(0, 3, 1, 2)
    return t.float().div(255) if x.dtype == np.uint8 else t.float()


def _resize(t: torch.Tensor, h: int, w: int) -> torch.Tensor:
    if tuple(t.shape[-2:]) == (h, w):
        return t
    return F.interpolate(t, size=(h, w), mode="bilinear", align_corners=False)


def _gray(img):
    return (0.2989 * img[:, 0:1] + 0.587 * img[:, 1:2] + 0.114 * img[:, 2:3])


def _adjust_hue(img, factor):
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    maxc, minc = img.max(1).values, img.min(1).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    h = (h + factor) % 1.0
    v = maxc
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p, q, t = (v * (1.0 - s)).clamp(0, 1), (v * (1.0 - s * f)).clamp(0, 1), (v * (1.0 - s * (1.0 - f))).clamp(0, 1)
    sel = torch.stack([torch.stack([v, q, p, p, t, v], 1), torch.stack([t, v, v, q, p, p], 1),
                       torch.stack([p, p, t, v, v, q], 1)], 1)  # (T, 3, 6, H, W)
    idx = i[:, None, None].expand(-1, 3, 1, -1, -1).long()
    return sel.gather(2, idx).squeeze(2)


def random_color_jitter(brightness, contrast, saturation, hue):
    """get_random_color_jitter (robonet_dataset.py:545-572): factors drawn with `random.uniform` in this order, the
    four adjustments applied in a `random.shuffle`d order; returns a function of a (T,3,H,W) clip."""
    ops = []
    bf = random.uniform(*brightness)
    ops.append(lambda im: (im * bf).clamp(0, 1))
    cf = random.uniform(*contrast)
    ops.append(lambda im: (cf * im + (1 - cf) * _gray(im).mean((1, 2, 3), keepdim=True)).clamp(0, 1))
    sf = random.uniform(*saturation)
    ops.append(lambda im: (sf * im + (1 - sf) * _gray(im)).clamp(0, 1))
    hf = random.uniform(*hue)
    ops.append(lambda im: _adjust_hue(im, hf))
    random.shuffle(ops)

    def apply(im):
        for op in ops:
            im = op(im)
        return im
    return apply


# --------------------------------------------------------------------------- #
# numeric preprocessing: one table entry per robot viewpoint, whole-trajectory array ops
# --------------------------------------------------------------------------- #
# The reference spreads this over per-item methods with per-robot branches (robonet_dataset.py:173-256,300-432).  Here
# the branches are DATA: a `Viewpoint` says where a robot's bounds come from and how its stored end-effector positions
# become metric world positions; `NumericPipeline` then maps the (T, .) arrays of a whole window at once.  The
# floating-point expressions keep the reference's order of operations in float32 (tests/golden/dataset_item.npz pins
# them bit for bit).
WORKSPACE_BOX = (np.array([0.015, -0.3, 0.1, 0, 0], dtype=np.float32),      # robonet_dataset.py:203-207: the planner's
                 np.array([0.55, 0.3, 0.4, 1, 1], dtype=np.float32))         # workspace for the authors' own robots
FRANKA_TO_LOCOBOT_XY = np.array([-0.365, -0.06103333])                       # LOCO_FRANKA_DIFF, robonet_dataset.py:21
FRANKA_PUSH_HEIGHT = 0.14


class Viewpoint:
    """What distinguishes one robot viewpoint's numeric preprocessing (robonet_dataset.py:200-208,305-322)."""
    __slots__ = ("name", "file_bounds", "stored_metric", "shift_xy", "fixed_z", "default_robot")

    def __init__(self, name: str):
        self.name = name
        own_rig = "locobot" in name or "franka" in name
        self.file_bounds = not own_rig       # RoboNet files carry low_bound / high_bound; the authors' rigs use the box
        self.stored_metric = own_rig         # RoboNet stores NORMALISED eef positions, the authors' rigs metric ones
        franka = "franka" in name and "locobot" not in name
        self.shift_xy = FRANKA_TO_LOCOBOT_XY if franka else None
        self.fixed_z = FRANKA_PUSH_HEIGHT if franka else None
        self.default_robot = "locobot" if "locobot" in name else ("franka" if "franka" in name else None)

    def bounds(self, traj):
        if self.file_bounds:
            return traj["low_bound"][:], traj["high_bound"][:]
        return WORKSPACE_BOX[0].copy(), WORKSPACE_BOX[1].copy()

    def metric_eef(self, states, low, high):
        """(T, 3) end-effector positions in the metric world frame, as a view / copy of states[:, :3]."""
        eef = states[:, :3]
        if not self.stored_metric:
            return denormalize(eef, low[:3], high[:3])
        if self.shift_xy is not None:
            eef[:, :2] += self.shift_xy
            eef[:, 2] = self.fixed_z
        return eef


def _homogeneous(points):
    """(n, 3) -> (4, n) homogeneous columns (float64, as np.ones promotes in the reference)."""
    return np.concatenate([points, np.ones((points.shape[0], 1))], 1).T


def _apply(matrix, points):
    return (matrix @ _homogeneous(points)).T[:, :3]


def fit_width(x, width):
    """Zero-pad the last axis of a (T, d) array up to `width` (robonet_dataset.py:209-229)."""
    if x.shape[-1] == width:
        return x
    assert width > x.shape[-1]
    return np.pad(x, [(0, 0), (0, width - x.shape[-1])])


def imputed_gripper_actions(actions, next_gripper, g_low, g_high):
    """`--impute_autograsp_action`: append the gripper command implied by the NEXT state's gripper reading, open / closed
    around the midpoint of its bounds (robonet_dataset.py:181-190), for all steps at once."""
    mid = (g_high + g_low) / 2.0
    cmd = np.where(next_gripper > mid, g_high, g_low).astype(np.float64)[:, None]
    return np.concatenate((actions, cmd), axis=-1)


class NumericPipeline:
    """Bounds, states and actions of one viewpoint under one `--preprocess_action` strategy, for whole windows."""

    def __init__(self, viewpoint: str, strategy: str):
        self.view = Viewpoint(viewpoint)
        self.strategy = strategy
        self.camera = "camera" in strategy
        self.w2c = self.c2w = None
        if self.camera:
            w2c, c2w = _camera_dicts()
            self.w2c, self.c2w = w2c[viewpoint], c2w[viewpoint]

    def bounds(self, raw_low, raw_high):
        """Camera strategies normalise in the camera frame: the box's 8 corners are transformed and re-boxed
        (robonet_dataset.py:231-255)."""
        low, high = raw_low.copy(), raw_high.copy()
        if self.camera:
            corners = np.array([[x, y, z] for x in (low[0], high[0]) for y in (low[1], high[1]) for z in (low[2], high[2])])
            cam = _apply(self.w2c, corners)
            low[:3], high[:3] = cam.min(0), cam.max(0)
        return low, high

    def states(self, states, low, high):
        """Stored states -> states normalised by (low, high), positions in the strategy's frame."""
        out = states.copy()
        eef = self.view.metric_eef(out, low, high)
        if self.camera:
            eef = _apply(self.w2c, eef)
        out[:, :3] = normalize(eef, low[:3], high[:3])
        out[:, 4] = normalize(out[:, 4], low[4], high[4])
        return out

    def actions(self, norm_states, actions, low, high):
        if self.strategy == "raw":
            return torch.from_numpy(actions)
        if self.strategy != "camera_raw":
            raise NotImplementedError(self.strategy)  # state_infer / camera_state_infer: as the reference
        # displacement of the eef in the camera frame.  The reference zeroes the recorded actions first
        # (robonet_dataset.py:383-384), so the "next" position equals the current one: kept, it is its arithmetic.
        out = np.zeros_like(actions)
        world = _apply(self.c2w, denormalize(norm_states[:, :3], low[:3], high[:3]))[:-1]
        out[:, :3] = _apply(self.w2c, world + out[:, :3]) - _apply(self.w2c, world)
        return torch.from_numpy(out)


class ImagePipeline:
    """uint8 frames / float masks of a window -> (T, C, h, w) tensors (robonet_dataset.py:257-300): ToTensor + bilinear
    resize of the whole window in one call; with augmentation one crop and one colour jitter per trajectory."""

    def __init__(self, height: int, width: int, augment: bool):
        self.h, self.w, self.augment = height, width, augment

    def __call__(self, frames, masks):
        video = _resize(_to_tensor(frames), self.h, self.w)
        mask = _resize(_to_tensor(masks), self.h, self.w)
        if self.augment:
            shrink = random.randint(0, 5)
            th, tw = self.h - shrink, self.w - shrink
            top = left = 0
            if shrink:  # tf.RandomCrop.get_params draws the corner from the torch generator, row first
                top = int(torch.randint(0, self.h - th + 1, size=(1,)).item())
                left = int(torch.randint(0, self.w - tw + 1, size=(1,)).item())
            jitter = random_color_jitter((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))
            window = (slice(None), slice(None), slice(top, top + th), slice(left, left + tw))
            video = jitter(_resize(video[window], self.h, self.w))
            mask = _resize(mask[window], self.h, self.w)
        return video, mask.type(torch.bool).type(torch.float32)


class RoboNetDataset(data.Dataset):
    """Trajectory files -> the reference's item dict (robonet_dataset.py:23-171): images (T,3,h,w) in [0,1], states (T,R)
    normalised, actions (T-1,A), masks (T,1,h,w) in {0,1}, qpos (T,Q), robot, folder, file_path, idx (+ low / high /
    raw_* for finetune_* experiments, high_movement with --load_movement_info)."""

    def __init__(self, hdf5_list, robot_list, config, augment_img=False, load_snippet=False, rng_seed=None):
        self._traj_names, self._traj_robots = hdf5_list, robot_list
        self._config = config
        self._data_root = config.data_root
        self._video_length = (config.n_past + config.n_future) if load_snippet else config.video_length
        self._action_dim = config.action_dim
        self._impute = getattr(config, "impute_autograsp_action", False)
        self._images = ImagePipeline(config.image_height, config.image_width, augment_img)
        self._pipelines = {}
        # window starts: the reference's RandomState(config.seed); data-parallel ranks pass their own seed
        self._rng = np.random.RandomState(config.seed if rng_seed is None else rng_seed)
        self._movement = self._movement_tables() if config.load_movement_info else None
        self._memory = {}
        if getattr(config, "preload_ram", False):
            self._memory = {i: self[i] for i in range(len(self))}

    def __len__(self):
        return len(self._traj_names)

    def _movement_tables(self):
        """folder -> {file path: high-movement flag} (obj_movement.pkl next to the trajectories, :36-48)."""
        tables = {}
        for folder_path in {os.path.dirname(n) for n in self._traj_names}:
            with open(os.path.join(folder_path, "obj_movement.pkl"), "rb") as f:
                tables[os.path.basename(folder_path)] = pickle.load(f)
        return tables

    def pipeline(self, viewpoint: str) -> NumericPipeline:
        pipe = self._pipelines.get(viewpoint)
        if pipe is None:
            pipe = self._pipelines[viewpoint] = NumericPipeline(viewpoint, self._config.preprocess_action)
        return pipe

    def _window(self, length: int, path: str):
        assert length >= self._video_length, f"{length}, {path}"
        if length == self._video_length:
            return 0, length
        start = self._rng.randint(0, length - self._video_length + 1)
        return start, start + self._video_length

    def read_actions(self, traj, start, end, g_low, g_high):
        """(end - start, A) float32 actions of the window; files with one dimension less get the imputed gripper
        command when `--impute_autograsp_action` is set."""
        actions = traj["actions"][:].astype(np.float32)
        have = actions.shape[1]
        if have != self._action_dim:
            if not (self._impute and have + 1 == self._action_dim):
                raise ValueError(f"file adim {have}, target adim {self._action_dim}")
            actions = imputed_gripper_actions(actions, traj["states"][:][1:, -1], g_low, g_high).astype(np.float32)
        return actions[start:end]

    def __getitem__(self, idx):
        if idx in self._memory:
            return self._memory[idx]
        cf = self._config
        name, viewpoint = self._traj_names[idx], self._traj_robots[idx]
        path = os.path.join(self._data_root, name)
        pipe = self.pipeline(viewpoint)
        with open_trajectory(path) as traj:
            frames_key = "observations" if "observations" in traj else "frames"
            assert frames_key in traj, path
            masks_key = "masks" if "masks" in traj else "mask"
            s, e = self._window(traj[frames_key].shape[0], path)
            frames = traj[frames_key][s:e]
            masks = traj[masks_key][s:e].astype(np.float32)
            raw_low, raw_high = pipe.view.bounds(traj)
            raw_states = fit_width(traj["states"][s:e].astype(np.float32), cf.robot_dim)
            raw_actions = self.read_actions(traj, s, e - 1, raw_low[4], raw_high[4])
            qpos = fit_width(traj["qpos"][s:e].astype(np.float32), cf.robot_joint_dim)
            robot = traj.attrs["robot"] if "robot" in traj.attrs else pipe.view.default_robot
            robot = robot.decode() if isinstance(robot, bytes) else str(robot)
        if not len(frames) == len(raw_states) == len(raw_actions) + 1 == len(masks):
            raise AssertionError(f"{path}, {frames.shape}, {raw_states.shape}, {raw_actions.shape}, {masks.shape}")
        if getattr(cf, "model_use_heatmap", False):
            raise NotImplementedError  # as the reference (robonet_dataset.py:130-133)
        low, high = pipe.bounds(raw_low, raw_high)
        images, masks = self._images(frames, masks)
        states = pipe.states(raw_states, low, high)
        actions = pipe.actions(states, raw_actions, low, high)
        folder = os.path.basename(os.path.dirname(name))
        item = {"images": images, "states": states, "actions": actions, "masks": masks, "robot": robot, "folder": folder,
                "file_path": path, "idx": idx, "qpos": qpos}
        if "finetune" in cf.experiment:  # the robot model of finetune_* windows needs the bounds (:148-166)
            item["low"], item["high"] = low, high
            if cf.preprocess_action != "raw":
                if not pipe.camera:
                    raise NotImplementedError
                world = raw_states.copy()  # world-frame states, normalised by the file's own bounds
                world[:, :3] = normalize(world[:, :3], raw_low[:3], raw_high[:3])
                world[:, 4] = normalize(world[:, 4], raw_low[4], raw_high[4])
                item.update(raw_low=raw_low, raw_high=raw_high, raw_actions=raw_actions.copy(), raw_states=world)
        if self._movement is not None:
            item["high_movement"] = self._movement[folder][path]
        return item


# --------------------------------------------------------------------------- #
# loaders (robonet_dataloaders.py:21-80, 137-199)
# --------------------------------------------------------------------------- #
def _robot_files(config, subdir, train_dirs, label):
    files, labels = [], []
    data_path = os.path.join(config.data_root, subdir)
    if not os.path.isdir(data_path):
        return []
    for folder in os.scandir(data_path):
        if folder.is_dir() and folder.name in train_dirs:
            for d in os.scandir(folder.path):
                if d.is_file() and d.path.lower().endswith(TRAJ_EXTENSIONS):
                    files.append(d.path)
                    labels.append(f"{label}_{folder.name}")
    fl = sorted(zip(files, labels), key=lambda x: x[0])
    random.seed(config.seed)
    random.shuffle(fl)
    return fl


def split_files(config):
    """The reference's discovery + shuffle + `train_test_split` (non-stratified): (X_train, X_test, y_train, y_test)."""
    from sklearn.model_selection import train_test_split
    fl = (_robot_files(config, "baxter_views", BAXTER_TRAIN_DIRS, "baxter")
          + _robot_files(config, "widowx_views", WIDOWX_TRAIN_DIRS, "widowx")
          + _robot_files(config, "sawyer_views", SAWYER_TRAIN_DIRS, "sawyer"))
    random.seed(config.seed)
    random.shuffle(fl)
    files, labels = [x[0] for x in fl], [x[1] for x in fl]
    return train_test_split(files, labels, test_size=1 - config.train_val_split,
                            random_state=np.random.RandomState(config.seed))


def collate(items):
    """Default collation of the item dicts (tensors / arrays stacked, strings listed), images already float."""
    return data.default_collate(items)


def _dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def create_loaders(config):
    """robonet_dataloaders.py:21-80.  Under torch.distributed (one process per GPU) the TRAIN loader shards the files
    with a DistributedSampler (same seed on every rank: disjoint shards of one shuffle, reshuffled per epoch through
    `set_epoch`, which `get_batch` calls) and each rank draws its own window starts; otherwise the ranks would train
    on identical batches and the gradient all-reduce would average copies of one gradient."""
    X_train, X_test, y_train, y_test = split_files(config)
    world, rank = _dist_info()
    train_data = RoboNetDataset(X_train, y_train, config, augment_img=getattr(config, "img_augmentation", False),
                                rng_seed=config.seed + rank if world > 1 else None)
    test_data = RoboNetDataset(X_test, y_test, config)
    common = dict(num_workers=config.data_threads, drop_last=False, pin_memory=True,
                  persistent_workers=config.data_threads > 0)
    if world > 1:
        sampler = data.distributed.DistributedSampler(train_data, num_replicas=world, rank=rank, shuffle=True,
                                                      seed=config.seed)
        train = data.DataLoader(train_data, batch_size=config.batch_size, sampler=sampler, **common)
    else:
        train = data.DataLoader(train_data, batch_size=config.batch_size, shuffle=True,
                                generator=torch.Generator().manual_seed(config.seed), **common)
    test = data.DataLoader(test_data, batch_size=config.test_batch_size, shuffle=True,
                           generator=torch.Generator().manual_seed(config.seed), **common)
    return train, test


LOCOBOT_FOLDERS = ["c0", "c1", "c2", "c3"]  # locobot_singleview_dataloader.py:11


def _locobot_files(config):
    """<data_root>/locobot_views/c{0..3}/*.hdf5 (or .npz), sorted then shuffled with the config seed; labels in discovery
    order, as the reference pairs them (locobot_singleview_dataloader.py:63-75)."""
    files, labels = [], []
    for folder in LOCOBOT_FOLDERS:
        path = os.path.join(config.data_root, "locobot_views", folder)
        if not os.path.isdir(path):
            continue
        for d in os.scandir(path):
            if d.is_file() and d.path.lower().endswith(TRAJ_EXTENSIONS):
                files.append(d.path)
                labels.append("locobot_" + folder)
    files = sorted(files)
    random.seed(config.seed)
    random.shuffle(files)
    return files, labels


def _plain_loader(ds, config, batch_size):
    return data.DataLoader(ds, num_workers=config.data_threads, batch_size=batch_size, shuffle=True, drop_last=False,
                           pin_memory=True, generator=torch.Generator().manual_seed(config.seed),
                           persistent_workers=config.data_threads > 0)


def create_transfer_loader(config, n_files: int = 400):
    """Zero-shot evaluation set of `--experiment train_robonet`: up to 400 trajectories of the unseen locobot
    (locobot_singleview_dataloader.py:59-92).  None when the data root has no locobot_views folder."""
    files, labels = _locobot_files(config)
    if not files:
        return None
    ds = RoboNetDataset(files[:n_files], labels[:n_files], config, augment_img=getattr(config, "img_augmentation", False))
    return _plain_loader(ds, config, config.batch_size)


def create_finetune_loaders(config):
    """`--experiment finetune_locobot`: finetune_num_test files for testing, the next finetune_num_train for training
    (locobot_singleview_dataloader.py:12-57)."""
    files, labels = _locobot_files(config)
    n_test, n_train = config.finetune_num_test, config.finetune_num_train
    train = RoboNetDataset(files[n_test:n_test + n_train], labels[n_test:n_test + n_train], config,
                           augment_img=getattr(config, "img_augmentation", False))
    test = RoboNetDataset(files[:n_test], labels[:n_test], config)
    return _plain_loader(train, config, config.batch_size), _plain_loader(test, config, config.test_batch_size)


def _start_epoch(loader, epoch: int):
    sampler = getattr(loader, "sampler", None)
    if hasattr(sampler, "set_epoch"):
        sampler.set_epoch(epoch)


# --------------------------------------------------------------------------- #
# host -> device
# --------------------------------------------------------------------------- #
def process_batch(data: dict, device) -> dict:
    """Changes tensor idx from batch-first to time-first and moves it to `device` (robonet_dataset.py:434-451)."""
    for k in TRANSPOSE_KEYS:
        if k in data:
            data[k] = data[k].transpose_(1, 0).to(device, non_blocking=True)
    return data


class DevicePrefetcher:
    """Iterates a loader one batch ahead: batch k+1 is copied (pinned memory -> HBM) on a side stream and transposed to
    time-first ON THE DEVICE while step k computes; `next()` makes the compute stream wait for that stream's event
    only.  A background thread keeps the (CPU-bound) loader iterator off the trainer's critical path."""

    def __init__(self, loader, device, depth: int = 2):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.q: Queue = Queue(maxsize=depth)
        self._stop = False
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _upload(self, batch):
        out = dict(batch)
        with torch.cuda.stream(self.stream):
            for k in TRANSPOSE_KEYS:
                if k in out:
                    t = out[k]
                    t = t if t.is_pinned() else t.pin_memory()
                    out[k] = t.to(self.device, non_blocking=True).transpose(0, 1).contiguous()
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def _run(self):
        torch.cuda.set_device(self.device)
        try:
            epoch = 0
            while not self._stop:
                _start_epoch(self.loader, epoch)
                epoch += 1
                for batch in self.loader:
                    if self._stop:
                        return
                    self.q.put(self._upload(batch))
        except BaseException as e:  # surfaced by the consumer
            self.q.put(e)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if isinstance(item, BaseException):
            raise item
        batch, ev = item
        torch.cuda.current_stream(self.device).wait_event(ev)
        for k in TRANSPOSE_KEYS:
            if k in batch:
                batch[k].record_stream(torch.cuda.current_stream(self.device))
        return batch

    def close(self):
        self._stop = True
        while not self.q.empty():
            self.q.get_nowait()


def get_batch(loader, device, prefetch: bool = True):
    """Infinite batch generator for a dataloader (robonet_dataset.py:454-467); `prefetch`: through DevicePrefetcher."""
    if prefetch and torch.device(device).type == "cuda":
        yield from DevicePrefetcher(loader, device)
        return
    epoch = 0
    while True:
        _start_epoch(loader, epoch)
        epoch += 1
        for batch in loader:
            yield process_batch(batch, device)


class SyntheticVideoDataset(data.Dataset):
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
