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


ELITE_SLOTS = [17, 923, 401, 655, 88, 760, 333, 512]  # where the graded candidates sit among the 1000


def demo_problem(ra: bool, N: int, T: int, seed: int = 6):
    """A planning problem whose elites are well separated (K / K+1 cost gap >= 1e-3): candidate N of the
    draw is the DEMONSTRATION -- the planner's per-step goal images are to be the model's own rollout of it (how the
    reference is driven: DemoGoalState.imgs holds one goal frame per step, trajectory_sampler.py:154) -- and the
    candidates at ELITE_SLOTS (taken modulo N) are graded blends demo + 0.1 k (candidate - demo), k = 1..8 (with the
    demo's robot states / masks).  Returns (problem with N + 1 action rows, the demonstration's actions)."""
    prob = synth_cem_problem(seed=seed, N=N + 1, T=T, with_robot=ra, goal_blend=0.15)
    acts = prob["actions"]
    demo = acts[N].clone()
    for k, j in enumerate(ELITE_SLOTS, start=1):
        j = j % N
        acts[j] = demo + 0.1 * k * (acts[j] - demo)
        if ra:
            prob["states"][:, j] = prob["states"][:, N]
            prob["masks"][:, j] = prob["masks"][:, N]
    return prob, demo


def frames_to_goal_images(obs) -> list:
    """(T, 3, H, W) float frames in [0, 1] -> T uint8 (H, W, 3) goal images, as a camera would hand them over."""
    return [np.clip(np.rint(np.asarray(o).transpose(1, 2, 0) * 255), 0, 255).astype(np.uint8) for o in obs]


def synth_arm_atlas(nx: int = 108, ny: int = 121, H: int = 64, W: int = 64, x_range=(0.015, 0.55),
                    y_range=(-0.3, 0.3), device="cpu") -> dict:
    """A synthetic robot-mask atlas for `robot_atlas.AtlasRobotModel` (5 mm grid over the workspace of
    src/cem/trajectory_sampler.py:22-23): node (j, i) holds the mask of an "arm" drawn from the image's bottom centre to
    the end effector at (x_i, y_j) seen by a fixed top-down camera -- a disc of radius 6 px plus a 3 px thick link.
    Stands in for the MuJoCo renders of the analytical robot models, which this image cannot produce."""
    xs = torch.linspace(x_range[0], x_range[1], nx, dtype=torch.float64)
    ys = torch.linspace(y_range[0], y_range[1], ny, dtype=torch.float64)
    m = arm_mask(xs.view(1, nx).expand(ny, nx).reshape(-1), ys.view(ny, 1).expand(ny, nx).reshape(-1), H, W, x_range,
                 y_range, device).view(ny, nx, H, W)
    return {"atlas": m, "x0": float(xs[0]), "y0": float(ys[0]), "dx": float(xs[1] - xs[0]), "dy": float(ys[1] - ys[0])}


def arm_mask(x, y, H: int = 64, W: int = 64, x_range=(0.015, 0.55), y_range=(-0.3, 0.3), device="cpu") -> torch.Tensor:
    """uint8 (n, H, W) masks of the synthetic arm with its end effector at the positions (x[k], y[k]): the "exact
    render" that `synth_arm_atlas` samples on a grid."""
    x = torch.as_tensor(x, dtype=torch.float64).to(device)
    y = torch.as_tensor(y, dtype=torch.float64).to(device)
    u = ((y - y_range[0]) / (y_range[1] - y_range[0]) * (W - 1)).float().view(-1, 1, 1)   # pixel column
    v = ((x_range[1] - x) / (x_range[1] - x_range[0]) * (H - 1)).float().view(-1, 1, 1)   # pixel row
    yy = torch.arange(H, device=device, dtype=torch.float32).view(1, H, 1)
    xx = torch.arange(W, device=device, dtype=torch.float32).view(1, 1, W)
    out = []
    bx, by = (W - 1) / 2, float(H - 1)
    for lo in range(0, u.shape[0], 2048):
        uu, vv = u[lo:lo + 2048], v[lo:lo + 2048]
        m = (xx - uu) ** 2 + (yy - vv) ** 2 <= 36
        dx, dy = uu - bx, vv - by
        t = (((xx - bx) * dx + (yy - by) * dy) / (dx * dx + dy * dy).clamp_min(1e-6)).clamp(0, 1)
        m |= (xx - (bx + t * dx)) ** 2 + (yy - (by + t * dy)) ** 2 <= 9
        out.append(m.to(torch.uint8))
    return torch.cat(out, 0)


def synth_state_dict(model, seed: int = 0, action_gain: float = 200.0, action_dim: int = 5) -> dict:
    """A deterministic random state_dict for `model` (name-keyed Philox streams, so every host and every rank produces
    the same bytes) whose activations stay O(1) through the 19 vgg layers and whose predictions DEPEND on the actions:
    conv weights He-scaled (gate convs 1/sqrt(fan_in), mu / logvar heads half of that), the action channels of the
    prior / frame-predictor input convs amplified by `action_gain`, BatchNorm affine and running statistics near (1, 0).
    With nn-style N(0, 0.02) weights (base.py:26-36) an untrained model's frames sit at sigmoid(~0) and candidates'
    costs differ by less than fp32 resolves: nothing a parity check or an elite selection could be judged on."""
    import math
    import zlib
    out = {}
    for key, ref in model.state_dict().items():
        shape = tuple(ref.shape)
        rng = np.random.Generator(np.random.Philox(key=[zlib.crc32(key.encode()), seed]))
        normal = lambda std, mean=0.0: torch.from_numpy(
            (rng.standard_normal(shape, dtype=np.float32) * np.float32(std) + np.float32(mean)).astype(np.float32))
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros((), dtype=torch.int64)
        elif key.endswith("running_mean"):
            out[key] = normal(0.1)
        elif key.endswith("running_var"):
            out[key] = torch.from_numpy(rng.uniform(0.5, 1.5, shape).astype(np.float32))
        elif len(shape) == 4:
            transposed = key == "decoder.upc5.1.weight"  # ConvTranspose2d stores (cin, cout, k, k)
            fan_in = (shape[0] if transposed else shape[1]) * shape[2] * shape[3]
            std = math.sqrt(2.0 / fan_in)
            if "gates" in key:
                std = math.sqrt(1.0 / fan_in)
            if "mu_net" in key or "logvar_net" in key:
                std = 0.5 * math.sqrt(1.0 / fan_in)
            w = normal(std)
            if key in ("prior_input_conv.weight", "frame_pred_input_conv.weight"):
                w[:, :action_dim] *= action_gain
            out[key] = w
        elif ".main.1." in key or "_norm." in key or ("gates.1." in key):  # BatchNorm / GroupNorm affine
            out[key] = normal(0.1, 1.0 if key.endswith("weight") else 0.0)
        else:  # conv biases
            out[key] = normal(0.05)
    return out


class SyntheticArmModel:
    """The analytical robot models' `predict_batch` contract (src/dataset/wx250s/wx250s_model.py:121-163) over the
    synthetic arm: states by the same propagation as `AtlasRobotModel` (pinned to the reference's goldens), masks
    rendered EXACTLY at every candidate's end-effector position (`arm_mask`) -- what an atlas of these masks
    approximates by its nearest grid node.  Stands in for the MuJoCo renders in tests and the benchmark."""

    def __init__(self, device, push_height: float = 0.12, H: int = 64, W: int = 64, frame_diff=(0.0, 0.0)):
        from .robot_atlas import WORKSPACE_HIGH, WORKSPACE_LOW, AtlasRobotModel
        self.device, self.H, self.W = torch.device(device), H, W
        self.x_range, self.y_range = (WORKSPACE_LOW[0], WORKSPACE_HIGH[0]), (WORKSPACE_LOW[1], WORKSPACE_HIGH[1])
        self._states = AtlasRobotModel(torch.zeros((1, 1, H, W), dtype=torch.uint8), 0.0, 0.0, 1.0, 1.0, push_height,
                                       frame_diff, device)
        self.calls = 0

    def atlas(self, nx: int, ny: int):
        """AtlasRobotModel over a (ny, nx) grid of this arm's renders, remembering this model as its `exact` source."""
        from .robot_atlas import AtlasRobotModel
        a = synth_arm_atlas(nx, ny, self.H, self.W, self.x_range, self.y_range, self.device)
        return AtlasRobotModel(a["atlas"], a["x0"], a["y0"], a["dx"], a["dy"], self._states.push_height,
                               self._states.diff, self.device, exact=self)

    def predict_batch(self, data, thick=True):
        self.calls += 1
        states, _ = self._states.predict_batch(data, thick)
        T1, N, _ = states.shape
        low = torch.as_tensor(data["low"]).to(self.device, torch.float32).reshape(-1, 5)[:, :2]
        high = torch.as_tensor(data["high"]).to(self.device, torch.float32).reshape(-1, 5)[:, :2]
        xy = states[..., :2] * (high - low) + low  # metric end-effector positions, (T+1, N, 2)
        masks = arm_mask(xy[..., 0].reshape(-1), xy[..., 1].reshape(-1), self.H, self.W, self.x_range, self.y_range,
                         self.device)
        return states, masks.view(T1, N, 1, self.H, self.W).float()
