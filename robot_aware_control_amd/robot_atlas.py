"""Robot-aware CEM inputs on the device: every candidate's robot states and masks in one launch.

With the robot-aware flags on (`model_use_mask`, `model_use_robot_state`, `dontcare` costs) the reference asks its
analytical robot model for the robot state and mask of every candidate at every step
(src/cem/trajectory_sampler.py:86-109 -> `predict_batch`, src/dataset/wx250s/wx250s_model.py:121-163,
src/dataset/locobot/locobot_model.py:100-140): a Python loop over the N candidates, each running inverse kinematics and
a MuJoCo segmentation render per step on the CPU -- seconds per CEM iteration, against ~1 s for the model rollouts of
1000 candidates here.

What that model computes depends on very little:
  * the state is the end effector moved by the planar action at a fixed push height (closed form);
  * the mask is the render of the IK solution at (x, y, push_height) with fixed pitch / roll -- a function of the
    end-effector POSITION only.
So the masks are rendered ONCE, by the wrapped analytical model, on a regular grid of end-effector positions over the
workspace (the "atlas", cached on disk) and a HIP kernel (`rac_cem_robot_inputs`) propagates the states of all
candidates and gathers their masks from the atlas: no per-candidate Python, IK or render in the planning loop.
The states reproduce the reference's arithmetic (golden vectors from its own `predict_batch`); a mask is exact when the
end effector sits on a grid node and otherwise that of the nearest node (grid spacing of a few millimetres is well
under a pixel of the 48x64 / 64x64 frames).

`AtlasRobotModel` has the analytical models' `predict_batch(data, thick)` contract, so it plugs into
`TrajectorySampler(robot_model=...)`, `CEMPolicy(robot_model=...)` and `PredictionTrainer.robot_model`.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

LOCO_WX250S_DIFF = (-0.13, -0.01)       # src/utils/camera_calibration.py:177
LOCO_FRANKA_DIFF = (-0.365, -0.06103333)  # :176
WORKSPACE_LOW = (0.015, -0.3, 0.1, 0.0, 0.0)   # src/cem/trajectory_sampler.py:22-23
WORKSPACE_HIGH = (0.55, 0.3, 0.4, 1.0, 1.0)


class AtlasRobotModel:
    def __init__(self, atlas: torch.Tensor, x0: float, y0: float, dx: float, dy: float, push_height: float,
                 frame_diff=(0.0, 0.0), device=None, preprocess_action: str = "raw", exact=None):
        """atlas: uint8 / bool (ny, nx, H, W), node (i, j) rendered with the end effector at (x0 + i dx, y0 + j dy) in
        the frame of the (de-normalised) states handed to `predict_batch`; `frame_diff`: offset of the robot's own
        frame, subtracted before and added after the propagation as the reference does (LOCO_WX250S_DIFF, ...).
        `preprocess_action`: the config's strategy -- anything but "raw" makes `predict_batch` read the world-frame
        `raw_actions / raw_states / raw_low / raw_high` of the batch, as the analytical models do
        (locobot_model.py:113-127).  `exact`: the model the atlas was rendered from (kept by `build`): the planner can
        re-roll its final elites with exactly rendered masks (TrajectorySampler, `cem_exact_elites`)."""
        dev = torch.device(device if device is not None else "cuda")
        if dev.type != "cuda":
            raise _lib.RacError("AtlasRobotModel runs on the GPU (no CPU fallback)")
        self.atlas = atlas.to(dev, torch.uint8).contiguous()
        self.ny, self.nx, self.H, self.W = self.atlas.shape
        self.x0, self.y0, self.dx, self.dy = float(x0), float(y0), float(dx), float(dy)
        self.push_height = float(push_height)
        self.diff = (float(frame_diff[0]), float(frame_diff[1]))
        self.device = dev
        self.preprocess_action = preprocess_action
        self.exact = exact
        self.shared_start_mask = None  # set by predict_batch: do all samples of the last batch share row 0?

    # ------------------------------------------------------------------ build / cache
    @classmethod
    def build(cls, robot_model, start_qpos, x_range, y_range, nx: int, ny: int, push_height: float,
              frame_diff=(0.0, 0.0), thick=True, low=WORKSPACE_LOW, high=WORKSPACE_HIGH, chunk=256, device=None):
        """Render the atlas with `robot_model` (the reference's WX250sAnalyticalModel / LocobotAnalyticalModel /
        FrankaAnalyticalModel, or anything with their `predict_batch`): one 1-step trajectory per grid node, from a
        fixed start to the node; the mask of step 1 is the node's mask."""
        xs = np.linspace(x_range[0], x_range[1], nx, dtype=np.float64)
        ys = np.linspace(y_range[0], y_range[1], ny, dtype=np.float64)
        low_t, high_t = torch.tensor([low], dtype=torch.float32), torch.tensor([high], dtype=torch.float32)
        start_xy = np.array([xs[nx // 2], ys[ny // 2]])
        start = torch.tensor([[start_xy[0], start_xy[1], push_height, 0.0, 0.0]], dtype=torch.float32)
        start_n = (start - low_t) / (high_t - low_t)
        nodes = np.stack(np.meshgrid(xs, ys, indexing="xy"), -1).reshape(-1, 2)  # row j * nx + i = (xs[i], ys[j])
        tiles = []
        qpos0 = torch.as_tensor(start_qpos, dtype=torch.float32)
        for lo in range(0, len(nodes), chunk):
            m = min(chunk, len(nodes) - lo)
            states = torch.zeros((2, m, 5))
            states[0] = start_n
            qpos = torch.zeros((2, m, qpos0.numel()))
            qpos[0] = qpos0
            actions = torch.zeros((1, m, 5))
            actions[0, :, :2] = torch.from_numpy(nodes[lo:lo + m] - start_xy).float()
            data = {"states": states, "qpos": qpos, "actions": actions, "low": low_t.repeat(m, 1), "high": high_t.repeat(m, 1)}
            _, masks = robot_model.predict_batch(data, thick=thick)
            tiles.append((masks[1, :, 0] != 0).to(torch.uint8).cpu())
        atlas = torch.cat(tiles, 0)
        atlas = atlas.view(ny, nx, atlas.shape[-2], atlas.shape[-1])
        return cls(atlas, xs[0], ys[0], (xs[-1] - xs[0]) / max(nx - 1, 1), (ys[-1] - ys[0]) / max(ny - 1, 1), push_height,
                   frame_diff, device, exact=robot_model)

    def save(self, path):
        np.savez_compressed(path, atlas=np.packbits(self.atlas.cpu().numpy(), axis=-1), shape=np.array(self.atlas.shape),
                            grid=np.array([self.x0, self.y0, self.dx, self.dy, self.push_height, *self.diff]))

    @classmethod
    def load(cls, path, device=None):
        z = np.load(path)
        shape = tuple(int(v) for v in z["shape"])
        atlas = np.unpackbits(z["atlas"], axis=-1)[..., :shape[-1]].reshape(shape)
        g = z["grid"]
        return cls(torch.from_numpy(atlas), g[0], g[1], g[2], g[3], g[4], (g[5], g[6]), device)

    # ------------------------------------------------------------------ the analytical models' contract
    def predict_batch(self, data, thick=True):
        """data: states (T+1, N, 5) whose row 0 holds each sample's normalised start state, actions (T, N, A) world-frame
        displacements, low / high (N, 5) -- with a non-"raw" `preprocess_action` the batch's `raw_*` entries instead
        (world-frame actions and states, the file's own bounds).  Every sample is propagated from ITS start with ITS
        bounds (a planner's candidates share them; the windows of a training batch do not).  Returns device tensors
        (states (T+1, N, 5) normalised, masks (T+1, N, 1, H, W) in {0, 1})."""
        dev = self.device
        pre = "" if self.preprocess_action == "raw" else "raw_"
        if pre and any(pre + k not in data for k in ("actions", "states", "low", "high")):
            raise KeyError(f"preprocess_action={self.preprocess_action!r}: the batch must carry raw_actions, raw_states, "
                           f"raw_low and raw_high (robonet_dataset.py:148-166)")
        src_states, src_actions = data[pre + "states"], data[pre + "actions"]
        T1, N, _ = src_states.shape
        T = T1 - 1
        actions = src_actions.to(dev, torch.float32).contiguous()
        if T and tuple(actions.shape[:2]) != (T, N):
            raise ValueError(f"actions {tuple(actions.shape)} do not match states {tuple(src_states.shape)}")
        A = actions.shape[2] if T else 2
        rows = lambda t: torch.as_tensor(t).to(dev, torch.float32).reshape(-1, 5).expand(N, 5).contiguous()
        s0 = torch.as_tensor(src_states[0]).reshape(-1, 5)
        if s0.device.type == "cpu":  # told to TrajectorySampler: every sample starts from one state -> one start mask
            self.shared_start_mask = bool((s0 == s0[:1]).all())
        else:
            self.shared_start_mask = None
        start, low, high = rows(src_states[0]), rows(data[pre + "low"]), rows(data[pre + "high"])
        states = torch.empty((T1, N, 5), device=dev, dtype=torch.float32)
        masks = torch.empty((T1, N, 1, self.H, self.W), device=dev, dtype=torch.float32)
        if not T:
            actions = torch.zeros((1, N, 2), device=dev)
        _lib.call("rac_cem_robot_inputs", actions.data_ptr(), start.data_ptr(), low.data_ptr(), high.data_ptr(),
                  self.atlas.data_ptr(), self.nx, self.ny, self.x0, self.y0, self.dx, self.dy, self.diff[0], self.diff[1],
                  self.push_height, states.data_ptr(), masks.data_ptr(), T, N, A, self.H * self.W, 1, _lib.stream_ptr())
        return states, masks
