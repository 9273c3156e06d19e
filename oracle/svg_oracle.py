"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the SVG / CEM hot path.

A from-scratch, *functional* PyTorch-CPU fp32 restatement of the reference
algorithm (no nn.Module hierarchy: a flat name->tensor dict keyed exactly like
the reference `state_dict()`).  Every function cites the reference file:line it
follows (paths relative to the reference root).

Pinned against the reference itself: `oracle/gen_golden.py` imports the real
reference in the build container, loads the same name-keyed synthetic weights
and dumps `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this file
against those vectors.  The product package never imports this module.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
TRACE = None  # debugging aid: set to a list to collect (name, tensor-with-retained-grad) of every vgg block output


class Forcing:
    """Test aid for the LeakyReLU / max-pool *selection* pattern of a training pass.

    `record=True`: collect, per vgg block prefix and call, the sign of the block output (LeakyReLU keeps the sign of
    its pre-activation) and the pre-activation itself; per pooling call the arg-max index map.
    `masks` / `pools` given: the pass takes the GIVEN selections instead of its own -- LeakyReLU slope 1 where the
    mask is set, 0.2 elsewhere; pooling reads the given positions.  Two fp32 implementations of the same network
    agree to ~1e-6, so a handful of pre-activations per million land on the other side of zero; with the selections
    of one implementation forced onto the other, every remaining operation is smooth and the gradients must agree
    to fp32 rounding.  `reps[prefix-root]`: calls that share one selection (the reference encodes the current frame
    twice per training step, dynamics.py:584,619)."""

    def __init__(self, masks=None, pools=None, rows_per_call=0, reps=None, record=False, l1_signs=None):
        self.masks, self.pools, self.rows, self.record = masks, pools, rows_per_call, record
        self.l1_signs, self.seen_l1 = l1_signs, []
        self.reps = reps or {"encoder": 2}
        self.calls = {}
        self.seen_sign, self.seen_pool = {}, {}
        self.flips = {}  # prefix -> [selections that differ, elements, max |pre-activation| / rms at a differing one]

    def _slice(self, table, key):
        c = self.calls.get(key, 0)
        self.calls[key] = c + 1
        step = c // self.reps.get(key.split(".")[0], 1)
        return table[key][step * self.rows:(step + 1) * self.rows]

    def leaky(self, prefix, y):
        if self.masks is None:
            if self.record:
                self.seen_sign.setdefault(prefix, []).append((y.detach() > 0))
            return F.leaky_relu(y, 0.2)
        m = self._slice(self.masks, prefix)
        if self.record:  # where does the forced selection differ from this pass's own, and how close to zero is it there
            yd = y.detach()
            diff = (yd > 0) != m
            n = int(diff.sum())
            worst = float(yd[diff].abs().max() / yd.pow(2).mean().sqrt()) if n else 0.0
            st = self.flips.setdefault(prefix, [0, 0, 0.0])
            st[0] += n
            st[1] += yd.numel()
            st[2] = max(st[2], worst)
        return torch.where(m, y, 0.2 * y)

    def abs(self, d):
        """|d| of the L1 reconstruction losses: d * sign with the sign pattern given per call (`l1_signs`, a list in
        call order of bool tensors `d > 0`), so that d |d| / dd agrees with the recorded pass also where d is zero to
        rounding."""
        if self.l1_signs is None:
            if self.record:
                self.seen_l1.append(d.detach() > 0)
            return d.abs()
        c = self.calls.get("l1", 0)
        self.calls["l1"] = c + 1
        pos = self.l1_signs[c]
        if self.record:
            dd = d.detach()
            diff = ((dd > 0) != pos) & (dd != 0)
            n = int(diff.sum())
            st = self.flips.setdefault("l1_sign", [0, 0, 0.0])
            st[0] += n
            st[1] += dd.numel()
            st[2] = max(st[2], float(dd[diff].abs().max()) if n else 0.0)
        return torch.where(pos, d, -d)

    def pool(self, tag, x):
        if self.pools is None:
            out, idx = F.max_pool2d(x, 2, 2, return_indices=True)
            if self.record:
                self.seen_pool.setdefault(tag, []).append(idx)
            return out
        idx = self._slice(self.pools, tag)
        b, c, h, w = x.shape
        out = x.flatten(2).gather(2, idx.flatten(2)).view(b, c, h // 2, w // 2)
        if self.record:  # a forced position that is not this pass's maximum: how far below the maximum is it
            own = F.max_pool2d(x.detach(), 2, 2)
            gap = (own - out.detach())
            n = int((gap > 0).sum())
            st = self.flips.setdefault(tag, [0, 0, 0.0])
            st[0] += n
            st[1] += gap.numel()
            st[2] = max(st[2], float(gap.max() / x.detach().pow(2).mean().sqrt()) if n else 0.0)
        return out


FORCING: Optional["Forcing"] = None


# --------------------------------------------------------------------------- #
# configuration
# --------------------------------------------------------------------------- #
@dataclass
class Cfg:
    """The flags of src/config/__init__.py:151-249 that shape the hot path."""

    image_width: int = 64
    image_height: int = 64
    channels: int = 3
    g_dim: int = 64
    z_dim: int = 16
    action_dim: int = 5
    robot_dim: int = 5
    batch_size: int = 2
    n_past: int = 1
    n_future: int = 2
    model_use_mask: bool = False
    model_use_future_mask: bool = False
    model_use_robot_state: bool = False
    model_use_future_robot_state: bool = False
    model_use_heatmap: bool = False
    model_use_future_heatmap: bool = False
    black_robot_input: bool = False
    last_frame_skip: bool = True
    reconstruction_loss: str = "l1"
    robot_pixel_weight: float = 0.0
    beta: float = 1e-4
    lr: float = 3e-4
    beta1: float = 0.9
    sample_mean: bool = False
    lstm_group_norm: bool = False
    # CEM / cost flags (src/config/__init__.py:315-357)
    candidates_batch_size: int = 200
    sparse_cost: bool = False
    reward_type: str = "dense"
    robot_cost_weight: float = 0.0
    world_cost_weight: float = 1.0
    topk: int = 5

    def enc_in_channels(self) -> int:  # dynamics.py:476-485
        c = self.channels
        if self.model_use_mask:
            c += 1
            if self.model_use_future_mask:
                c += 1
        if self.model_use_heatmap:
            c += 1
            if self.model_use_future_heatmap:
                c += 1
        return c

    def lstm_in_channels(self) -> int:  # dynamics.py:490-494
        c = self.g_dim + self.action_dim + self.z_dim
        if self.model_use_robot_state:
            c += self.robot_dim
        if self.model_use_future_robot_state:
            c += self.robot_dim
        return c

    def post_in_channels(self) -> int:  # dynamics.py:504-507
        return self.g_dim + (self.robot_dim if self.model_use_robot_state else 0)

    def prior_in_channels(self) -> int:  # dynamics.py:505-510
        c = self.g_dim + self.action_dim
        if self.model_use_robot_state:
            c += self.robot_dim
        if self.model_use_future_robot_state:
            c += self.robot_dim
        return c


def cfg_from_namespace(ns) -> Cfg:
    """Pick the Cfg fields out of any argparse-style namespace."""
    kw = {}
    for f in Cfg.__dataclass_fields__:
        if hasattr(ns, f):
            kw[f] = getattr(ns, f)
    return Cfg(**kw)


# --------------------------------------------------------------------------- #
# parameter inventory (== reference state_dict(), dynamics.py:468-534)
# --------------------------------------------------------------------------- #
ENC_PLAN = (("c1", (None, 64, 64)), ("c2", (64, 128, 128)), ("c3", (128, 256, 256, 256)),
            ("c4", (256, 512, 512, None)))
DEC_PLAN = (("upc2", (None, 512, 512, 256)), ("upc3", (512, 256, 256, 128)),
            ("upc4", (256, 128, 64)), ("upc5", (128, 64)))


def _vgg_entries(prefix: str, cin: int, cout: int):
    base = f"{prefix}.main"
    return [
        (f"{base}.0.weight", (cout, cin, 3, 3), "conv_w"),
        (f"{base}.1.weight", (cout,), "bn_w"),
        (f"{base}.1.bias", (cout,), "bn_b"),
        (f"{base}.1.running_mean", (cout,), "bn_rm"),
        (f"{base}.1.running_var", (cout,), "bn_rv"),
        (f"{base}.1.num_batches_tracked", (), "bn_nbt"),
    ]


def _lstm_entries(prefix: str, g: int, group_norm: bool = False):
    out = []
    for i, k in enumerate((5, 3)):  # lstm.py:207-212
        base = f"{prefix}.lstm.{i}"
        if group_norm:  # NormConvLSTMCell (lstm.py:163-172)
            for part in ("ih_gates", "hh_gates"):
                out.append((f"{base}.{part}.0.weight", (4 * g, g, k, k), "conv_w"))
                out.append((f"{base}.{part}.0.bias", (4 * g,), "conv_b"))
                out.append((f"{base}.{part}.1.weight", (4 * g,), "bn_w"))
                out.append((f"{base}.{part}.1.bias", (4 * g,), "bn_b"))
            out.append((f"{base}.c_norm.weight", (g,), "bn_w"))
            out.append((f"{base}.c_norm.bias", (g,), "bn_b"))
        else:
            out.append((f"{base}.gates.weight", (4 * g, 2 * g, k, k), "conv_w"))
            out.append((f"{base}.gates.bias", (4 * g,), "conv_b"))
    return out


def param_spec(cfg: Cfg) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) in reference state_dict order."""
    g, z = cfg.g_dim, cfg.z_dim
    spec = []
    for name, chans in ENC_PLAN:  # vgg_64.py:99-119
        chans = [cfg.enc_in_channels() if c is None and i == 0 else (g if c is None else c)
                 for i, c in enumerate(chans)]
        for i in range(len(chans) - 1):
            spec += _vgg_entries(f"encoder.{name}.{i}", chans[i], chans[i + 1])
    spec.append(("frame_pred_input_conv.weight", (g, cfg.lstm_in_channels(), 3, 3), "conv_w"))
    spec.append(("frame_pred_input_conv.bias", (g,), "conv_b"))
    spec += _lstm_entries("frame_predictor", g, cfg.lstm_group_norm)
    spec.append(("posterior_input_conv.weight", (g, cfg.post_in_channels(), 3, 3), "conv_w"))
    spec.append(("posterior_input_conv.bias", (g,), "conv_b"))
    spec.append(("prior_input_conv.weight", (g, cfg.prior_in_channels(), 3, 3), "conv_w"))
    spec.append(("prior_input_conv.bias", (g,), "conv_b"))
    for nm in ("posterior", "prior"):
        spec += _lstm_entries(nm, g, cfg.lstm_group_norm)
        for head in ("mu_net", "logvar_net"):  # lstm.py:273-274
            spec.append((f"{nm}.{head}.weight", (z, g, 3, 3), "conv_w"))
            spec.append((f"{nm}.{head}.bias", (z,), "conv_b"))
    for name, chans in DEC_PLAN:  # vgg_64.py:208-219
        chans = [g if c is None else c for c in chans]
        for i in range(len(chans) - 1):
            spec += _vgg_entries(f"decoder.{name}.{i}", chans[i], chans[i + 1])
    spec.append(("decoder.upc5.1.weight", (64, cfg.channels + 1, 3, 3), "convT_w"))
    spec.append(("decoder.upc5.1.bias", (cfg.channels + 1,), "conv_b"))
    return spec


def is_buffer(kind: str) -> bool:
    return kind in ("bn_rm", "bn_rv", "bn_nbt")


def make_weights(cfg: Cfg, seed: int = 0, action_gain: float = 1.0,
                 randomize_bn_stats: bool = True) -> Dict[str, Tensor]:
    """Name-keyed, counter-based synthetic state_dict (SURVEY.md section 8c golden plan).

    Every tensor depends only on (seed, key, shape): Philox keyed by crc32(key).
    Conv weights are He-scaled so activations stay O(1) through 19 vgg layers;
    LSTM gate weights 1/sqrt(fan_in); `action_gain` scales the action-channel
    slices of the prior / frame-predictor input convs.
    """
    out: Dict[str, Tensor] = {}
    for key, shape, kind in param_spec(cfg):
        rng = np.random.Generator(np.random.Philox(key=[zlib.crc32(key.encode()), seed]))
        if kind in ("conv_w", "convT_w"):
            fan_in = int(np.prod(shape[1:])) if kind == "conv_w" else shape[0] * shape[2] * shape[3]
            std = math.sqrt(1.0 / fan_in) if "gates." in key else math.sqrt(2.0 / fan_in)
            if "mu_net" in key or "logvar_net" in key:
                std = 0.5 * math.sqrt(1.0 / fan_in)
            w = rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
            if key in ("prior_input_conv.weight", "frame_pred_input_conv.weight") and action_gain != 1.0:
                w[:, : cfg.action_dim] *= np.float32(action_gain)
            out[key] = torch.from_numpy(w)
        elif kind == "conv_b":
            out[key] = torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05))
        elif kind == "bn_w":
            out[key] = torch.from_numpy(1 + rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1))
        elif kind == "bn_b":
            out[key] = torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1))
        elif kind == "bn_rm":
            v = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
            out[key] = torch.from_numpy(v if randomize_bn_stats else np.zeros(shape, np.float32))
        elif kind == "bn_rv":
            v = rng.uniform(0.5, 1.5, shape).astype(np.float32)
            out[key] = torch.from_numpy(v if randomize_bn_stats else np.ones(shape, np.float32))
        elif kind == "bn_nbt":
            out[key] = torch.zeros((), dtype=torch.int64)
        else:  # pragma: no cover
            raise KeyError(kind)
    return out


# --------------------------------------------------------------------------- #
# model pieces
# --------------------------------------------------------------------------- #
def vgg_block(sd: Dict[str, Tensor], prefix: str, x: Tensor, training: bool) -> Tensor:
    """Conv3x3(no bias) -> BatchNorm2d -> LeakyReLU(0.2)   (vgg_64.py:8-18)."""
    base = f"{prefix}.main"
    y = F.conv2d(x, sd[f"{base}.0.weight"], None, 1, 1)
    if training:
        sd[f"{base}.1.num_batches_tracked"] += 1
    y = F.batch_norm(y, sd[f"{base}.1.running_mean"], sd[f"{base}.1.running_var"],
                     sd[f"{base}.1.weight"], sd[f"{base}.1.bias"], training, BN_MOMENTUM, BN_EPS)
    y = F.leaky_relu(y, 0.2) if FORCING is None else FORCING.leaky(prefix, y)
    if TRACE is not None and y.requires_grad:
        y.retain_grad()
        TRACE.append((prefix, y))
    return y


def _stack(sd, prefix, n, x, training):
    for i in range(n):
        x = vgg_block(sd, f"{prefix}.{i}", x, training)
    return x


def encoder(sd: Dict[str, Tensor], x: Tensor, training: bool):
    """ConvEncoder.forward (vgg_64.py:122-129): returns (h4, [h1,h2,h3,h4])."""
    pool = (lambda tag, t: F.max_pool2d(t, 2, 2)) if FORCING is None else FORCING.pool
    h1 = _stack(sd, "encoder.c1", 2, x, training)
    h2 = _stack(sd, "encoder.c2", 2, pool("encoder.pool1", h1), training)
    h3 = _stack(sd, "encoder.c3", 3, pool("encoder.pool2", h2), training)
    h4 = _stack(sd, "encoder.c4", 3, pool("encoder.pool3", h3), training)
    return h4, [h1, h2, h3, h4]


def _up2(x: Tensor) -> Tensor:
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)  # nearest x2


def decoder(sd: Dict[str, Tensor], vec: Tensor, skip: List[Tensor], training: bool) -> Tensor:
    """ConvDecoder.forward (vgg_64.py:223-241); skip[3] is unused."""
    d2 = _stack(sd, "decoder.upc2", 3, vec, training)
    d3 = _stack(sd, "decoder.upc3", 3, torch.cat([_up2(d2), skip[2]], 1), training)
    d4 = _stack(sd, "decoder.upc4", 2, torch.cat([_up2(d3), skip[1]], 1), training)
    d5 = vgg_block(sd, "decoder.upc5.0", torch.cat([_up2(d4), skip[0]], 1), training)
    y = F.conv_transpose2d(d5, sd["decoder.upc5.1.weight"], sd["decoder.upc5.1.bias"], 1, 1)
    return torch.sigmoid(y)


def convlstm_cell(sd, prefix: str, layer: int, x: Tensor, state: Tuple[Tensor, Tensor]):
    """ConvLSTMCell.forward (lstm.py:129-149); gate order in, remember, out, cell."""
    h_prev, c_prev = state
    pad = 2 if layer == 0 else 1
    gates = F.conv2d(torch.cat([x, h_prev], 1), sd[f"{prefix}.lstm.{layer}.gates.weight"],
                     sd[f"{prefix}.lstm.{layer}.gates.bias"], 1, pad)
    gi, gf, go, gg = gates.chunk(4, 1)
    c = torch.sigmoid(gf) * c_prev + torch.sigmoid(gi) * torch.tanh(gg)
    h = torch.sigmoid(go) * torch.tanh(c)
    return h, c


def norm_convlstm_cell(sd, prefix: str, layer: int, x: Tensor, state: Tuple[Tensor, Tensor]):
    """NormConvLSTMCell.forward (lstm.py:174-198)."""
    h_prev, c_prev = state
    pad = 2 if layer == 0 else 1
    base = f"{prefix}.lstm.{layer}"
    ih = F.group_norm(F.conv2d(x, sd[f"{base}.ih_gates.0.weight"], sd[f"{base}.ih_gates.0.bias"], 1, pad), 16,
                      sd[f"{base}.ih_gates.1.weight"], sd[f"{base}.ih_gates.1.bias"])
    hh = F.group_norm(F.conv2d(h_prev, sd[f"{base}.hh_gates.0.weight"], sd[f"{base}.hh_gates.0.bias"], 1, pad), 16,
                      sd[f"{base}.hh_gates.1.weight"], sd[f"{base}.hh_gates.1.bias"])
    gi, gf, go, gg = (ih + hh).chunk(4, 1)
    c = torch.sigmoid(gf) * c_prev + torch.sigmoid(gi) * torch.tanh(gg)
    c = F.group_norm(c, 16, sd[f"{base}.c_norm.weight"], sd[f"{base}.c_norm.bias"])
    return torch.sigmoid(go) * torch.tanh(c), c


def convlstm(sd, prefix: str, x: Tensor, hidden: List[Tuple[Tensor, Tensor]]) -> Tensor:
    """ConvLSTM.forward (lstm.py:252-257); mutates `hidden` in place."""
    cell = norm_convlstm_cell if f"{prefix}.lstm.0.c_norm.weight" in sd else convlstm_cell
    for i in range(2):
        hidden[i] = cell(sd, prefix, i, x, hidden[i])
        x = hidden[i][0]
    return x


def gaussian_convlstm(sd, prefix: str, x: Tensor, hidden, eps: Optional[Tensor]):
    """GaussianConvLSTM.forward (lstm.py:281-286); eps is the N(0,1) draw of lstm.py:278."""
    h = convlstm(sd, prefix, x, hidden)
    mu = F.conv2d(h, sd[f"{prefix}.mu_net.weight"], sd[f"{prefix}.mu_net.bias"], 1, 1)
    logvar = F.conv2d(h, sd[f"{prefix}.logvar_net.weight"], sd[f"{prefix}.logvar_net.bias"], 1, 1)
    if eps is None:
        eps = torch.randn_like(mu)
    z = eps * torch.exp(0.5 * logvar) + mu
    return z, mu, logvar


def init_hidden(cfg: Cfg, batch: int) -> Dict[str, List[Tuple[Tensor, Tensor]]]:
    """SVGConvModel.init_hidden (dynamics.py:536-542; lstm.py:218-250)."""
    h, w = cfg.image_height // 8, cfg.image_width // 8
    mk = lambda: [(torch.zeros(batch, cfg.g_dim, h, w), torch.zeros(batch, cfg.g_dim, h, w)) for _ in range(2)]
    return {"frame_predictor": mk(), "posterior": mk(), "prior": mk()}


def _tile(v: Tensor, h: int, w: int) -> Tensor:
    return v[:, :, None, None].expand(-1, -1, h, w)  # dynamics.py:592


def svg_forward(sd, cfg: Cfg, hidden, image, mask, robot, heatmap, action,
                next_image=None, next_mask=None, next_robot=None, next_heatmap=None,
                skip=None, force_use_prior=False, sample_mean=False, training=False,
                eps_prior: Optional[Tensor] = None, eps_post: Optional[Tensor] = None):
    """SVGConvModel.forward (dynamics.py:544-644), including the posterior quirk
    that encodes the *current* frame again (dynamics.py:619)."""
    img = image
    if cfg.model_use_heatmap:
        img = torch.cat([img, heatmap], 1)
    if cfg.model_use_mask:
        img = torch.cat([img, mask], 1)
    h, curr_skip = encoder(sd, img, training)
    if cfg.last_frame_skip or skip is None:
        skip = curr_skip
    hh, ww = cfg.image_height // 8, cfg.image_width // 8
    a = _tile(action, hh, ww)
    mu = logvar = None
    cond = [a]
    if cfg.model_use_robot_state:
        if cfg.model_use_future_robot_state:
            r, r_next = robot
            cond += [_tile(r, hh, ww), _tile(r_next, hh, ww)]
        else:
            cond += [_tile(robot, hh, ww)]
    prior_in = F.conv2d(torch.cat(cond + [h], 1), sd["prior_input_conv.weight"], sd["prior_input_conv.bias"], 1, 1)
    z_p, mu_p, logvar_p = gaussian_convlstm(sd, "prior", prior_in, hidden["prior"], eps_prior)
    z = mu_p if sample_mean else z_p
    if next_image is not None:
        h_target = encoder(sd, img, training)[0]  # sic: `img`, not the next frame
        if cfg.model_use_robot_state:
            post_x = torch.cat([_tile(next_robot, hh, ww), h_target], 1)
        else:
            post_x = h_target
        post_in = F.conv2d(post_x, sd["posterior_input_conv.weight"], sd["posterior_input_conv.bias"], 1, 1)
        z_t, mu, logvar = gaussian_convlstm(sd, "posterior", post_in, hidden["posterior"], eps_post)
        if not force_use_prior:
            z = z_t
    frame_in = F.conv2d(torch.cat(cond + [h, z], 1), sd["frame_pred_input_conv.weight"],
                        sd["frame_pred_input_conv.bias"], 1, 1)
    h_pred = convlstm(sd, "frame_predictor", frame_in, hidden["frame_predictor"])
    x_pred = decoder(sd, h_pred, skip, training)
    return x_pred, skip, mu, logvar, mu_p, logvar_p


# --------------------------------------------------------------------------- #
# helpers, losses, costs
# --------------------------------------------------------------------------- #
def zero_robot_region(mask: Tensor, image: Tensor) -> Tensor:
    """src/utils/image.py:5-19 (tensor branch, not in place)."""
    return torch.where(mask.bool().expand(-1, 3, -1, -1), image * 0, image)


def composite(x_pred4: Tensor, x_prev: Tensor) -> Tensor:
    """trainer.py:406-407 / trajectory_sampler.py:149-150."""
    m = x_pred4[:, 3:4]
    return (1 - m) * x_prev + m * x_pred4[:, :3]


def _abs(d: Tensor) -> Tensor:
    return d.abs() if FORCING is None else FORCING.abs(d)


def l1_loss(pred, target, batch_weight=None):
    """losses.py:13-19."""
    d = _abs(target - pred)
    return d.mean() if batch_weight is None else torch.mean(batch_weight * d.mean((1, 2, 3)))


def mse_loss(pred, target):
    """losses.py:11 (nn.MSELoss)."""
    return ((pred - target) ** 2).mean()


def _masked_diff(pred, target, mask, robot_weight):
    m3 = mask.bool().expand(-1, 3, -1, -1)
    d = target - pred
    d = torch.where(m3, d * robot_weight, d)
    n_world = (~m3).sum((1, 2, 3)) + 1
    return d, n_world


def dontcare_l1_loss(pred, target, mask, robot_weight, batch_weight=None):
    """losses.py:35-50."""
    d, n_world = _masked_diff(pred, target, mask, robot_weight)
    per = _abs(d).sum((1, 2, 3))
    if batch_weight is not None:
        per = batch_weight * per
    return torch.mean(per / n_world)


def dontcare_mse_loss(pred, target, mask, robot_weight):
    """losses.py:21-33."""
    d, n_world = _masked_diff(pred, target, mask, robot_weight)
    return torch.mean((d ** 2).sum((1, 2, 3)) / n_world)


def robot_mse(pred, target, mask):
    """losses.py:52-64."""
    m3 = mask.bool().expand(-1, 3, -1, -1)
    d = torch.where(m3, target - pred, torch.zeros(()))
    return torch.mean((d ** 2).sum((1, 2, 3)) / (m3.sum((1, 2, 3)) + 1))


def world_mse(pred, target, mask):
    """losses.py:66-78."""
    m3 = mask.bool().expand(-1, 3, -1, -1)
    d = torch.where(m3, torch.zeros(()), target - pred)
    return torch.mean((d ** 2).sum((1, 2, 3)) / ((~m3).sum((1, 2, 3)) + 1))


def kl_loss(mu1, logvar1, mu2, logvar2, bs):
    """losses.py:97-106."""
    s1, s2 = torch.exp(0.5 * logvar1), torch.exp(0.5 * logvar2)
    kld = torch.log(s2 / s1) + (torch.exp(logvar1) + (mu1 - mu2) ** 2) / (2 * torch.exp(logvar2)) - 0.5
    assert kld.shape[0] == bs
    return kld.sum() / bs


def recon_loss(cfg: Cfg, pred, target, mask=None, batch_weight=None):
    """PredictionTrainer._recon_loss (trainer.py:149-161)."""
    kind = cfg.reconstruction_loss
    if kind == "mse":
        return mse_loss(pred, target)
    if kind == "l1":
        return l1_loss(pred, target, batch_weight)
    if kind == "dontcare_mse":
        return dontcare_mse_loss(pred, target, mask, cfg.robot_pixel_weight)
    if kind == "dontcare_l1":
        return dontcare_l1_loss(pred, target, mask, cfg.robot_pixel_weight, batch_weight)
    raise NotImplementedError(kind)


def img_l2_cost(curr, goal):
    """ImgL2Cost._call_tensor (losses.py:224-235): -sqrt(sum (255*d)^2), fp32."""
    return -(((255 * (curr - goal)) ** 2).sum((1, 2, 3)).sqrt()).numpy()


def img_dontcare_cost(curr, goal, curr_mask, goal_mask):
    """ImgDontcareCost._call_tensor (losses.py:244-263)."""
    tot = curr_mask.bool() | goal_mask.bool()
    d = (255 * (curr - goal)) ** 2
    d = torch.where(tot.expand(-1, 3, -1, -1), torch.zeros(()), d)
    dist = d.sum((1, 2, 3)).sqrt() / (~tot).sum((1, 2, 3))
    return -dist.numpy()


# --------------------------------------------------------------------------- #
# train step (PredictionTrainer._train_step, trainer.py:326-465)
# --------------------------------------------------------------------------- #
@dataclass
class TrainState:
    sd: Dict[str, Tensor]
    cfg: Cfg
    optimizer: torch.optim.Optimizer = None
    param_keys: List[str] = field(default_factory=list)

    @staticmethod
    def create(cfg: Cfg, sd: Dict[str, Tensor]) -> "TrainState":
        sd = {k: v.clone() for k, v in sd.items()}
        keys = [k for k, _, kind in param_spec(cfg) if not is_buffer(kind)]
        for k in keys:
            sd[k].requires_grad_(True)
        opt = torch.optim.Adam([sd[k] for k in keys], lr=cfg.lr, betas=(cfg.beta1, 0.999))  # trainer.py:109-122
        return TrainState(sd, cfg, opt, keys)


def train_step(ts: TrainState, data: Dict[str, Tensor], eps: Optional[List[Tuple[Tensor, Tensor]]] = None,
               use_truth: Optional[List[bool]] = None, do_update: bool = True) -> Dict[str, float]:
    """One optimiser step.  `eps[i-1] = (eps_prior, eps_post)` for time index i;
    `use_truth[i]` replaces the scheduled-sampling coin of trainer.py:147."""
    cfg, sd = ts.cfg, ts.sd
    x, states, ac, mask = data["images"], data["states"], data["actions"], data["masks"]
    losses: Dict[str, float] = {"recon_loss": 0.0, "robot_loss": 0.0, "world_loss": 0.0, "kld": 0.0}
    for k in ts.param_keys:
        sd[k].grad = None
    bs = min(cfg.batch_size, x.shape[1])
    hidden = init_hidden(cfg, bs)
    dontcare = "dontcare" in cfg.reconstruction_loss or cfg.black_robot_input
    recon = 0
    kld = 0
    x_pred = None
    skip = None
    for i in range(1, cfg.n_past + cfg.n_future):
        truth = True if (i == 1 or use_truth is None) else use_truth[i]
        x_j = x[i - 1] if truth else x_pred.clone()
        m_j, r_j, a_j = mask[i - 1], states[i - 1], ac[i - 1]
        x_i, m_i, r_i = x[i], mask[i], states[i]
        x_j_black, x_i_black = x_j, x_i
        if dontcare:
            x_j_black, x_i_black = zero_robot_region(m_j, x_j), zero_robot_region(m_i, x_i)
        if cfg.last_frame_skip:
            skip = None
        m_in = torch.cat([m_j, m_i], 1) if cfg.model_use_future_mask else m_j
        r_in = (r_j, r_i) if cfg.model_use_future_robot_state else r_j
        m_next = m_i.repeat(1, 2, 1, 1) if cfg.model_use_future_mask else m_i
        e_p, e_q = eps[i - 1] if eps is not None else (None, None)
        out = svg_forward(sd, cfg, hidden, x_j_black, m_in, r_in, None, a_j, x_i_black, m_next, r_i, None,
                          skip, training=True, eps_prior=e_p, eps_post=e_q)
        x4, curr_skip, mu, logvar, mu_p, logvar_p = out
        x_pred = composite(x4, x_j)
        if i <= cfg.n_past:
            skip = curr_skip
        view = recon_loss(cfg, x_pred, x_i, m_i)
        recon = recon + view
        losses["recon_loss"] += view.item()
        with torch.no_grad():
            losses["robot_loss"] += robot_mse(x_pred, x_i, m_i).item()
            losses["world_loss"] += world_mse(x_pred, x_i, m_i).item()
        kl = kl_loss(mu, logvar, mu_p, logvar_p, bs)
        kld = kld + kl
        losses["kld"] += kl.item()
    loss = recon + kld * cfg.beta
    loss.backward()
    if do_update:
        ts.optimizer.step()
    return {k: v / cfg.n_future for k, v in losses.items()}


# --------------------------------------------------------------------------- #
# eval path (src/utils/metrics.py:13-78, PredictionTrainer._eval_step trainer.py:566-734)
# --------------------------------------------------------------------------- #
def psnr(estimates: Tensor, targets: Tensor) -> Tensor:
    """metrics.py:61-78: both mapped to (x+1)/2, per-sample mean over (C,H,W), 10 log10(1/mse)."""
    mse = (((estimates + 1) / 2) - ((targets + 1) / 2)).pow(2).mean((1, 2, 3))
    return 10 * torch.log(1.0 / mse) / math.log(10)


def ssim_map(img1: Tensor, img2: Tensor) -> Tensor:
    """metrics.py:13-58: 11x11 Gaussian (sigma 1.5) depthwise window, zero padding, C1=0.01^2, C2=0.03^2."""
    g = torch.tensor([math.exp(-(i - 5) ** 2 / (2 * 1.5 ** 2)) for i in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    w = (g @ g.t()).float().expand(3, 1, 11, 11).contiguous()
    blur = lambda t: F.conv2d(t, w, padding=5, groups=3)
    m1, m2 = blur(img1), blur(img2)
    v1, v2, v12 = blur(img1 * img1) - m1 * m1, blur(img2 * img2) - m2 * m2, blur(img1 * img2) - m1 * m2
    return ((2 * m1 * m2 + 1e-4) * (2 * v12 + 9e-4)) / ((m1 * m1 + m2 * m2 + 1e-4) * (v1 + v2 + 9e-4))


@torch.no_grad()
def eval_step(sd, cfg: Cfg, data: Dict[str, Tensor], n_eval: int, autoregressive: bool,
              eps: List[Tuple[Tensor, Tensor]]) -> Dict[str, float]:
    """_eval_step with the model in eval mode; `eps[i-1]` = (prior, posterior) draws of step i."""
    x, states, ac, masks = data["images"], data["states"], data["actions"], data["masks"]
    bs = x.shape[1]
    hidden = init_hidden(cfg, bs)
    prefix = "autoreg" if autoregressive else "1step"
    dontcare = "dontcare" in cfg.reconstruction_loss or cfg.black_robot_input
    losses: Dict[str, float] = {}
    k_losses: Dict[str, float] = {}
    add = lambda k, v: losses.__setitem__(k, losses.get(k, 0.0) + float(v))
    x_pred = skip = None
    for i in range(1, n_eval):
        x_j = x_pred.clone() if (autoregressive and i > 1) else x[i - 1]
        m_j, r_j, a_j, m_i, r_i, x_i = masks[i - 1], states[i - 1], ac[i - 1], masks[i], states[i], x[i]
        x_j_black, x_i_black = (zero_robot_region(m_j, x_j), zero_robot_region(m_i, x_i)) if dontcare else (x_j, x_i)
        if cfg.last_frame_skip:
            skip = None
        m_in = torch.cat([m_j, m_i], 1) if cfg.model_use_future_mask else m_j
        r_in = (r_j, r_i) if cfg.model_use_future_robot_state else r_j
        m_next = m_i.repeat(1, 2, 1, 1) if cfg.model_use_future_mask else m_i
        x4, curr_skip, mu, logvar, mu_p, logvar_p = svg_forward(
            sd, cfg, hidden, x_j_black, m_in, r_in, None, a_j, x_i_black, m_next, r_i, None, skip,
            force_use_prior=True, eps_prior=eps[i - 1][0], eps_post=eps[i - 1][1])
        x_pred = composite(x4, x_j)
        if i <= cfg.n_past:
            skip = curr_skip
        add(f"{prefix}_recon_loss", recon_loss(cfg, x_pred, x_i, m_i))
        add(f"{prefix}_robot_loss", robot_mse(x_pred, x_i, m_i))
        wm = float(world_mse(x_pred, x_i, m_i))
        add(f"{prefix}_world_loss", wm)
        pb, tb = zero_robot_region(m_i, x_pred), zero_robot_region(m_i, x_i)
        p = float(psnr(tb.clamp(0, 1), pb.clamp(0, 1)).mean())
        s_ = float(ssim_map(tb, pb).mean())
        add(f"{prefix}_psnr", p)
        add(f"{prefix}_ssim", s_)
        if autoregressive:
            k_losses.update({f"{i}_step_psnr": p, f"{i}_step_ssim": s_, f"{i}_step_world_loss": wm})
        add(f"{prefix}_kld", kl_loss(mu, logvar, mu_p, logvar_p, bs))
    out = {k: v / (n_eval - 1) for k, v in losses.items()}
    out.update(k_losses)
    return out


# --------------------------------------------------------------------------- #
# CEM (src/cem/trajectory_sampler.py:35-199, src/cem/cem.py:56-111)
# --------------------------------------------------------------------------- #
@torch.no_grad()
def cem_rollouts(sd, cfg: Cfg, action_sequences: Tensor, start_img: np.ndarray, goal_imgs: List[np.ndarray],
                 goal_masks: Optional[List[np.ndarray]] = None, states: Optional[Tensor] = None,
                 masks: Optional[Tensor] = None, opt_traj: Optional[Tensor] = None,
                 ret_obs: bool = False) -> Dict[str, np.ndarray]:
    """generate_model_rollouts.  `states (T+1,N,R)` / `masks (T+1,N,1,H,W)` stand in
    for `robot_model.predict_batch` (trajectory_sampler.py:86-109, CPU MuJoCo; out of scope)."""
    if opt_traj is not None:  # trajectory_sampler.py:62-68
        opt = torch.cat([opt_traj, torch.zeros((len(opt_traj), 3))], 1).unsqueeze(0)
        action_sequences = torch.cat([action_sequences, opt])
    N, T = action_sequences.shape[0], action_sequences.shape[1]
    per = cfg.candidates_batch_size
    nb = max(N // per, 1)
    sum_cost = np.zeros(N)
    g_imgs = torch.stack([torch.from_numpy(g).permute(2, 0, 1).float() / 255 for g in goal_imgs])
    g_masks = torch.stack([torch.from_numpy(g) for g in goal_masks]) if goal_masks is not None else None
    dontcare_in = "dontcare" in cfg.reconstruction_loss or cfg.black_robot_input
    obs = torch.zeros((N, T, 3, cfg.image_height, cfg.image_width)) if ret_obs else None
    for b in range(nb):
        s = b * per
        e = (b + 1) * per if b < nb - 1 else N
        n = e - s
        hidden = init_hidden(cfg, n)
        curr = (torch.from_numpy(start_img.copy()).permute(2, 0, 1).float() / 255).expand(n, -1, -1, -1)
        for t in range(T):
            ac = action_sequences[s:e, t]
            mask = masks[t, s:e] if cfg.model_use_mask else None
            state = states[t, s:e] if cfg.model_use_robot_state else None
            if dontcare_in:
                curr = zero_robot_region(masks[t, s:e], curr)
            if cfg.model_use_future_mask:
                mask = torch.cat([mask, masks[t + 1, s:e]], 1)
            if cfg.model_use_future_robot_state:
                state = (state, states[t + 1, s:e])
            x4 = svg_forward(sd, cfg, hidden, curr, mask, state, None, ac, sample_mean=cfg.sample_mean,
                             eps_prior=torch.zeros(n, cfg.z_dim, cfg.image_height // 8, cfg.image_width // 8)
                             if cfg.sample_mean else None)[0]
            nxt = composite(x4, curr)
            if dontcare_in:
                nxt = zero_robot_region(masks[t + 1, s:e], nxt)
            gi = t if t < len(g_imgs) else -1
            if not cfg.sparse_cost or t == T - 1:
                if cfg.world_cost_weight != 0:
                    if cfg.reward_type == "dontcare":
                        rew = cfg.world_cost_weight * img_dontcare_cost(nxt, g_imgs[gi], masks[t + 1, s:e], g_masks[gi])
                    else:
                        rew = cfg.world_cost_weight * img_l2_cost(nxt, g_imgs[gi])
                    sum_cost[s:e] += rew
            if ret_obs:
                obs[s:e, t] = nxt
            curr = nxt
    out: Dict[str, np.ndarray] = {}
    if opt_traj is not None:
        out["optimal_sum_cost"] = sum_cost[-1]
        sum_cost = sum_cost[:-1]
    out["sum_cost"] = sum_cost
    if ret_obs:
        out["obs_all"] = obs.numpy()
    return out


def cem_refit(act_seq: Tensor, sum_cost: np.ndarray, k: int):
    """cem.py:96-104: top-k (largest cost = best), unbiased std, floor 1e-3."""
    costs = torch.from_numpy(sum_cost)
    top_costs, top_idx = costs.topk(k)
    top = torch.index_select(act_seq, 0, top_idx)
    std, mean = torch.std_mean(top, dim=0)
    std = torch.max(0.001 * torch.ones_like(std), std)
    return mean, std, top_idx, top_costs


def cem_get_action(sd, cfg: Cfg, start_img, goal_imgs, goal_masks, horizon: int, opt_iter: int, n_cand: int,
                   topk: int, init_std: float, noise: List[Tensor], states=None, masks=None):
    """CEMPolicy.get_action (cem.py:56-111).  `noise[i]` is the N(0,1) draw (N,T-1,2)
    that `Normal(mean,std).sample((N,))` consumes at iteration i (sample = mean + std*noise)."""
    T, A = horizon, 2
    mean = torch.zeros(T - 1, A)
    std = torch.ones(T - 1, A) * init_std
    trace = []
    for i in range(opt_iter):
        act = mean + std * noise[i]
        if i == 0:
            act[-1] = 0
        act.clamp_(-0.05, 0.05)
        padded = torch.cat([act, torch.zeros((n_cand, T - 1, 3))], 2)
        ro = cem_rollouts(sd, cfg, padded, start_img, goal_imgs, goal_masks, states, masks)
        mean, std, top_idx, top_costs = cem_refit(act, ro["sum_cost"], topk)
        trace.append({"act_seq": act.numpy().copy(), "sum_cost": ro["sum_cost"].copy(),
                      "top_idx": top_idx.numpy().copy(), "mean": mean.numpy().copy(), "std": std.numpy().copy()})
    return mean.numpy(), trace
