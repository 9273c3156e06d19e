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


class RoboNetDataset(data.Dataset):
    """Same contract as the reference class (robonet_dataset.py:23-171): item = dict(images (T,3,h,w) in [0,1],
    states (T,R) normalised, actions (T-1,A), masks (T,1,h,w) in {0,1}, robot, folder, file_path, idx, qpos (T,Q))."""

    def __init__(self, hdf5_list, robot_list, config, augment_img=False, load_snippet=False):
        self._traj_names = hdf5_list
        self._traj_robots = robot_list
        self._config = config
        self._data_root = config.data_root
        if config.load_movement_info:  # robonet_dataset.py:36-48
            movement_vp = {}
            for name in self._traj_names:
                folder = os.path.basename(os.path.dirname(name))
                if folder in movement_vp:
                    continue
                with open(os.path.join(os.path.dirname(name), "obj_movement.pkl"), "rb") as f:
                    movement_vp[folder] = pickle.load(f)
            self._movement_vp = movement_vp
        self._video_length = config.video_length
        if load_snippet:
            self._video_length = config.n_past + config.n_future
        self._action_dim = config.action_dim
        self._impute_autograsp_action = getattr(config, "impute_autograsp_action", False)
        self._augment_img = augment_img
        self._rng = np.random.RandomState(config.seed)
        self._memory = {}
        if getattr(config, "preload_ram", False):
            for i in range(len(self._traj_names)):
                self._memory[i] = self.__getitem__(i)

    def __len__(self):
        return len(self._traj_names)

    def __getitem__(self, idx):
        if idx in self._memory:
            return self._memory[idx]
        cf = self._config
        name = self._traj_names[idx]
        robot_viewpoint = self._traj_robots[idx]
        path = os.path.join(self._data_root, name)
        with open_trajectory(path) as hf:
            assert "frames" in hf or "observations" in hf
            image_key = "observations" if "observations" in hf else "frames"
            mask_key = "masks" if "masks" in hf else "mask"
            ep_len = hf[image_key].shape[0]
            assert ep_len >= self._video_length, f"{ep_len}, {path}"
            start, end = 0, ep_len
            if ep_len > self._video_length:
                start = self._rng.randint(0, ep_len - self._video_length + 1)
                end = start + self._video_length
            images = hf[image_key][start:end]
            raw_low, raw_high = self._load_bounds(hf, robot_viewpoint)
            g_low, g_high = raw_low[4], raw_high[4]
            states = self._load_states(hf, start, end)
            actions = self._load_actions(hf, g_low, g_high, start, end - 1)
            if cf.preprocess_action != "raw":
                raw_states, raw_actions = states.copy(), actions.copy()
            masks = hf[mask_key][start:end].astype(np.float32)
            qpos = self._load_qpos(hf, start, end)
            assert len(images) == len(states) == len(actions) + 1 == len(masks), \
                f"{path}, {images.shape}, {states.shape}, {actions.shape}, {masks.shape}"
            low, high = self._preprocess_bounds(raw_low, raw_high, idx)
            images, masks = self._preprocess_images_masks(images, masks)
            states = self._preprocess_states(states, low, high, robot_viewpoint, idx)
            actions = self._preprocess_actions(states, actions, low, high, idx)
            if "robot" in hf.attrs:
                robot = hf.attrs["robot"]
                robot = robot.decode() if isinstance(robot, bytes) else str(robot)
            elif "locobot" in robot_viewpoint:
                robot = "locobot"
            elif "franka" in robot_viewpoint:
                robot = "franka"
            folder = os.path.basename(os.path.dirname(name))
            if getattr(cf, "model_use_heatmap", False):
                raise NotImplementedError  # as the reference (robonet_dataset.py:130-133)
        out = {"images": images, "states": states, "actions": actions, "masks": masks, "robot": robot,
               "folder": folder, "file_path": path, "idx": idx, "qpos": qpos}
        if "finetune" in cf.experiment:  # robonet_dataset.py:148-166
            out["low"], out["high"] = low, high
            if cf.preprocess_action == "raw":
                pass
            elif "camera" in cf.preprocess_action:
                out["raw_low"], out["raw_high"], out["raw_actions"] = raw_low, raw_high, raw_actions
                raw_states[:, :3] = normalize(raw_states[:, :3], raw_low[:3], raw_high[:3])
                raw_states[:, 4] = normalize(raw_states[:, 4], raw_low[4], raw_high[4])
                out["raw_states"] = raw_states
            else:
                raise NotImplementedError
        if cf.load_movement_info:
            out["high_movement"] = self._movement_vp[folder][path]
        return out

    # ---- loaders (robonet_dataset.py:173-229) ----
    def _load_actions(self, fp, gripper_low, gripper_high, start, end):
        actions = fp["actions"][:].astype(np.float32)
        a_T, adim = actions.shape[0], actions.shape[1]
        if self._action_dim == adim:
            return actions[start:end]
        if self._impute_autograsp_action and adim + 1 == self._action_dim:
            action_append = np.zeros((a_T, 1))
            next_state = fp["states"][:][1:, -1]
            high_val, low_val = gripper_high[-1], gripper_low[-1]
            midpoint = (high_val + low_val) / 2.0
            for t, s in enumerate(next_state):
                action_append[t, 0] = high_val if s > midpoint else low_val
            return np.concatenate((actions, action_append), axis=-1)[start:end].astype(np.float32)
        raise ValueError(f"file adim {adim}, target adim {self._action_dim}")

    def _load_bounds(self, fp, robot_viewpoint):
        if "locobot" in robot_viewpoint or "franka" in robot_viewpoint:
            return (np.array([0.015, -0.3, 0.1, 0, 0], dtype=np.float32),
                    np.array([0.55, 0.3, 0.4, 1, 1], dtype=np.float32))
        return fp["low_bound"][:], fp["high_bound"][:]

    def _pad_last(self, x, dim):
        if x.shape[-1] != dim:
            assert dim > x.shape[-1]
            x = np.pad(x, [(0, 0), (0, dim - x.shape[-1])])
        return x

    def _load_states(self, fp, start, end):
        return self._pad_last(fp["states"][start:end].astype(np.float32), self._config.robot_dim)

    def _load_qpos(self, fp, start, end):
        return self._pad_last(fp["qpos"][start:end].astype(np.float32), self._config.robot_joint_dim)

    # ---- preprocessing (robonet_dataset.py:231-356) ----
    def _preprocess_bounds(self, low, high, idx):
        low, high = low.copy(), high.copy()
        if "camera" in self._config.preprocess_action:
            world2cam = _camera_dicts()[0][self._traj_robots[idx]]
            xs, ys, zs = (low[0], high[0]), (low[1], high[1]), (low[2], high[2])
            box = np.array([[x, y, z, 1.0] for x in xs for y in ys for z in zs]).T  # the 8 corners, homogeneous
            c_box = ((world2cam @ box).T)[:, :3]
            low[:3], high[:3] = np.min(c_box, 0), np.max(c_box, 0)
        return low, high

    def _preprocess_images_masks(self, images, masks):
        cf = self._config
        h, w = cf.image_height, cf.image_width
        video = _resize(_to_tensor(images), h, w)
        mask = _resize(_to_tensor(masks), h, w)
        if not self._augment_img:
            return video, mask.type(torch.bool).type(torch.float32)
        # one random crop and one colour jitter per trajectory (robonet_dataset.py:261-291)
        rand_crop = random.randint(0, 5)
        th, tw = h - rand_crop, w - rand_crop
        if (th, tw) == (h, w):
            i = j = 0
        else:  # tf.RandomCrop.get_params draws i then j from the torch generator
            i = int(torch.randint(0, h - th + 1, size=(1,)).item())
            j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
        jitter = random_color_jitter((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))
        video = jitter(_resize(video[:, :, i:i + th, j:j + tw], h, w))
        mask = _resize(mask[:, :, i:i + th, j:j + tw], h, w).type(torch.bool).type(torch.float32)
        return video, mask

    def _preprocess_states(self, states, low, high, robot_viewpoint, idx):
        states = states.copy()
        if "locobot" in robot_viewpoint:
            eef_pos = states[:, :3]
        elif "franka" in robot_viewpoint:
            eef_pos = states[:, :3]
            eef_pos[:, :2] += np.array([-0.365, -0.06103333])  # LOCO_FRANKA_DIFF (robonet_dataset.py:21)
            eef_pos[:, 2] = 0.14
        else:
            eef_pos = denormalize(states[:, :3], low[:3], high[:3])
        if "camera" in self._config.preprocess_action:
            world2cam = _camera_dicts()[0][self._traj_robots[idx]]
            eef_pos = np.concatenate([eef_pos, np.ones((eef_pos.shape[0], 1))], 1).T
            eef_pos = ((world2cam @ eef_pos).T)[:, :3]
        states[:, :3] = normalize(eef_pos, low[:3], high[:3])
        states[:, 4] = normalize(states[:, 4], low[4], high[4])
        return states

    def _preprocess_actions(self, states, actions, low, high, idx):
        strategy = self._config.preprocess_action
        if strategy == "raw":
            return torch.from_numpy(actions)
        if strategy != "camera_raw":
            raise NotImplementedError  # state_infer / camera_state_infer: as the reference
        w_to_c, c_to_w = (d[self._traj_robots[idx]] for d in _camera_dicts())
        states = states.copy()
        actions = np.zeros_like(actions)  # sic (robonet_dataset.py:383-384): the recorded actions are discarded
        c_eef = denormalize(states[:, :3], low[:3], high[:3])
        c_eef = np.concatenate([c_eef, np.ones((c_eef.shape[0], 1))], 1).T
        eef = ((c_to_w @ c_eef).T)[:-1, :3]
        nxt = eef + actions[:, :3]
        hom = lambda p: np.concatenate([p, np.ones((p.shape[0], 1))], 1).T
        actions[:, :3] = ((w_to_c @ hom(nxt)).T)[:, :3] - ((w_to_c @ hom(eef)).T)[:, :3]
        return torch.from_numpy(actions)


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


def create_loaders(config):
    X_train, X_test, y_train, y_test = split_files(config)
    train_data = RoboNetDataset(X_train, y_train, config, augment_img=getattr(config, "img_augmentation", False))
    test_data = RoboNetDataset(X_test, y_test, config)
    mk = lambda ds, bs: data.DataLoader(ds, num_workers=config.data_threads, batch_size=bs, shuffle=True, drop_last=False,
                                        pin_memory=True, generator=torch.Generator().manual_seed(config.seed),
                                        persistent_workers=config.data_threads > 0)
    return mk(train_data, config.batch_size), mk(test_data, config.test_batch_size)


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
            while not self._stop:
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
    while True:
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
