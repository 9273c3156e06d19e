"""CPU tests of the data path (robot_aware_control_amd/data.py) against the reference's numerics (golden vectors
captured from `RoboNetDataset`'s own methods, oracle/gen_golden.py:gen_dataset) and its item / batch contract
(robonet_dataset.py:69-171,434-467; robonet_dataloaders.py:21-80)."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

from robot_aware_control_amd import data as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cfg(root, **kw):
    d = dict(data_root=root, load_movement_info=False, video_length=8, n_past=1, n_future=2, action_dim=4, robot_dim=5,
             robot_joint_dim=7, impute_autograsp_action=False, image_width=64, image_height=48, seed=3,
             preload_ram=False, preprocess_action="raw", experiment="train_robonet", model_use_heatmap=False,
             train_val_split=0.75, img_augmentation=False, data_threads=0, batch_size=3, test_batch_size=2)
    d.update(kw)
    return argparse.Namespace(**d)


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_synthetic_robonet as mk
    root = str(tmp_path_factory.mktemp("robonet"))
    assert mk.write(root, per_view=4, length=10, seed=1) == 16
    return root


def test_numeric_preprocessing_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "dataset_item.npz"))

    class FP(dict):
        attrs = {}
    fp = FP({k[3:]: g[k] for k in g.files if k.startswith("in_")})
    ds = object.__new__(D.RoboNetDataset)  # the window reader without files
    ds._action_dim, ds._impute = 4, False
    for tag, view in (("sawyer", "sawyer_sudri0_c0"), ("locobot", "locobot_c0"), ("franka", "franka_c0")):
        pipe = D.NumericPipeline(view, "raw")
        low, high = pipe.view.bounds(fp)
        states = D.fit_width(fp["states"][2:8].astype(np.float32), 5)
        actions = ds.read_actions(fp, 2, 7, low[4], high[4])
        qpos = D.fit_width(fp["qpos"][2:8].astype(np.float32), 7)
        plow, phigh = pipe.bounds(low, high)
        pstates = pipe.states(states, plow, phigh)
        pact = pipe.actions(pstates, actions, plow, phigh)
        for name, val in (("low", low), ("high", high), ("states", states), ("actions", actions), ("qpos", qpos),
                          ("pstates", pstates), ("pactions", pact.numpy())):
            ref = g[f"{tag}_{name}"]
            assert val.dtype == ref.dtype and np.array_equal(val, ref), (tag, name)
    # --impute_autograsp_action: the vectorised form of the reference's per-step loop (robonet_dataset.py:181-190)
    nxt = fp["states"][:][1:, -1]
    want = np.array([[1.0 if s_ > 0.5 else 0.0] for s_ in nxt])
    got = D.imputed_gripper_actions(fp["actions"][:].astype(np.float32), nxt, 0.0, 1.0)
    assert got.shape[1] == fp["actions"].shape[1] + 1 and np.array_equal(got[:, -1:], want)
    assert np.array_equal(D.normalize(g["norm_in"], fp["low_bound"], fp["high_bound"]), g["norm"])
    assert np.array_equal(D.denormalize(g["norm_in"], fp["low_bound"], fp["high_bound"]), g["denorm"])


def test_item_contract_and_image_semantics(tree):
    c = cfg(tree)
    Xtr, Xte, ytr, yte = D.split_files(c)
    assert len(Xtr) == 12 and len(Xte) == 4 and set(ytr + yte) <= {"sawyer_sudri0_c0", "sawyer_sudri2_c1",
                                                                   "widowx_widowx1_c0", "baxter_left_c0"}
    assert D.split_files(c)[0] == Xtr  # same seed, same split
    ds = D.RoboNetDataset(Xtr, ytr, c)
    it = ds[0]
    assert set(it) == {"images", "states", "actions", "masks", "robot", "folder", "file_path", "idx", "qpos"}
    assert it["images"].shape == (8, 3, 48, 64) and it["images"].dtype == torch.float32
    assert it["masks"].shape == (8, 1, 48, 64) and set(it["masks"].unique().tolist()) <= {0.0, 1.0}
    assert it["states"].shape == (8, 5) and it["actions"].shape == (7, 4) and it["qpos"].shape == (8, 7)
    assert it["robot"] in ("sawyer", "widowx", "baxter") and it["folder"] in it["file_path"]
    # the window start comes from RandomState(seed) exactly when the episode is longer than video_length
    z = np.load(it["file_path"])
    rng = np.random.RandomState(c.seed)
    T0 = z["frames"].shape[0]
    s = rng.randint(0, T0 - 8 + 1) if T0 > 8 else 0
    ref_img = torch.nn.functional.interpolate(torch.from_numpy(z["frames"][s:s + 8]).permute(0, 3, 1, 2).float() / 255,
                                              size=(48, 64), mode="bilinear", align_corners=False)
    assert torch.equal(it["images"], ref_img)
    ref_mask = torch.nn.functional.interpolate(torch.from_numpy(z["mask"][s:s + 8].astype(np.float32))[:, None],
                                               size=(48, 64), mode="bilinear", align_corners=False)
    assert torch.equal(it["masks"], (ref_mask != 0).float())  # any bilinear overlap with the robot -> robot
    low, high = z["low_bound"], z["high_bound"]
    st = z["states"][s:s + 8].copy()
    ref_states = st.copy()
    ref_states[:, :3] = D.normalize(D.denormalize(st[:, :3], low[:3], high[:3]), low[:3], high[:3])
    ref_states[:, 4] = D.normalize(st[:, 4], low[4], high[4])
    assert np.array_equal(it["states"], ref_states) and np.array_equal(it["actions"].numpy(), z["actions"][s:s + 7])
    # load_snippet: the window shrinks to n_past + n_future
    assert D.RoboNetDataset(Xte, yte, c, load_snippet=True)[1]["images"].shape[0] == 3
    # augmentation keeps the contract (one crop + one jitter per trajectory, binary masks)
    aug = D.RoboNetDataset(Xtr, ytr, c, augment_img=True)[2]
    assert aug["images"].shape == (8, 3, 48, 64) and float(aug["images"].min()) >= 0 and float(aug["images"].max()) <= 1
    assert set(aug["masks"].unique().tolist()) <= {0.0, 1.0}


def test_loaders_and_time_first_batches(tree):
    c = cfg(tree)
    train_loader, test_loader = D.create_loaders(c)
    assert len(train_loader.dataset) == 12 and len(test_loader.dataset) == 4
    gen = D.get_batch(train_loader, torch.device("cpu"), prefetch=False)
    seen = 0
    for _ in range(6):  # more than one epoch: the generator is infinite
        b = next(gen)
        B = b["images"].shape[1]
        assert b["images"].shape == (8, B, 3, 48, 64) and b["masks"].shape == (8, B, 1, 48, 64)
        assert b["states"].shape == (8, B, 5) and b["actions"].shape == (7, B, 4) and b["qpos"].shape == (8, B, 7)
        assert len(b["robot"]) == B and isinstance(b["robot"][0], str)
        seen += B
    assert seen > 12
    # process_batch is the reference's in-place transpose
    raw = next(iter(test_loader))
    img = raw["images"].clone()
    out = D.process_batch(raw, torch.device("cpu"))
    assert torch.equal(out["images"], img.transpose(0, 1))


def test_transfer_and_finetune_loaders(tmp_path):
    """The unseen-locobot loaders (locobot_singleview_dataloader.py:12-92): file discovery under locobot_views/c*, the
    seed-shuffled split, items through the fixed-workspace viewpoint (metric states, `observations` / `masks` keys)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_synthetic_robonet as mk
    root = str(tmp_path)
    mk.write(root, per_view=1, length=10, seed=2, locobot=3)
    c = cfg(root, robot_joint_dim=5, finetune_num_test=2, finetune_num_train=3)
    loader = D.create_transfer_loader(c)
    assert len(loader.dataset) == 6
    b = next(iter(loader))
    assert b["images"].shape[1:] == (8, 3, 48, 64) and b["qpos"].shape[-1] == 5 and set(b["robot"]) == {"locobot"}
    assert float(b["states"][..., :3].min()) >= 0 and float(b["states"][..., :3].max()) <= 1  # normalised by the workspace box
    it = loader.dataset[0]
    z = np.load(it["file_path"])
    low, high = D.WORKSPACE_BOX
    assert np.array_equal(it["states"][:, :3], D.normalize(z["states"][:8, :3] if len(z["states"]) == 8 else
                                                          it["states"][:, :3] * (high[:3] - low[:3]) + low[:3], low[:3], high[:3]))
    train, test = D.create_finetune_loaders(c)
    assert len(test.dataset) == 2 and len(train.dataset) == 3
    assert not set(test.dataset._traj_names) & set(train.dataset._traj_names)
    assert D.create_transfer_loader(cfg(str(tmp_path / "nothing_here"))) is None


def test_shim_falls_through_to_the_reference_tree(tmp_path, monkeypatch):
    """With this repo ahead of the reference on sys.path, modules that exist here shadow the reference's and the rest
    of `src.*` (the analytical robot models, camera calibration, mbrl loops) still resolves into the reference."""
    ref = tmp_path / "ref"
    for pkg in ("src", "src/dataset", "src/dataset/wx250s", "src/utils", "src/cem", "src/mbrl"):
        (ref / pkg).mkdir(parents=True)
        (ref / pkg / "__init__.py").write_text("")
    (ref / "src/dataset/wx250s/wx250s_model.py").write_text("WHO = 'reference'\n")
    (ref / "src/utils/camera_calibration.py").write_text("LOCO_WX250S_DIFF = 7\n")
    (ref / "src/cem/cem.py").write_text("WHO = 'reference'\n")
    (ref / "src/mbrl/runner.py").write_text("WHO = 'reference'\n")
    monkeypatch.syspath_prepend(str(ref))
    monkeypatch.syspath_prepend(ROOT)
    for m in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        monkeypatch.delitem(sys.modules, m)
    import importlib
    assert importlib.import_module("src.dataset.wx250s.wx250s_model").WHO == "reference"
    assert importlib.import_module("src.utils.camera_calibration").LOCO_WX250S_DIFF == 7
    assert importlib.import_module("src.mbrl.runner").WHO == "reference"
    assert hasattr(importlib.import_module("src.cem.cem"), "CEMPolicy")  # ours shadows the reference's
    assert importlib.import_module("src.dataset.robonet.robonet_dataset").RoboNetDataset is D.RoboNetDataset
