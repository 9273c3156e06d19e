"""SVGConvModel on librac_hip.so -- drop-in for the reference
`src.prediction.models.dynamics.SVGConvModel` (reference dynamics.py:457-644):
same constructor (`SVGConvModel(config)`), `init_hidden`, 12-argument `forward`,
and `state_dict()` keys / shapes, so reference checkpoints load unchanged.

Differences that are deliberate and invisible to callers:
  * feature maps live in HBM as NHWC; tensors handed back to the caller are
    logical NCHW views of them (channels_last strides);
  * all parameters are views into ONE flat fp32 buffer (and their gradients into
    another): the fused Adam step and the DDP all-reduce work on flat memory;
    conv weights are stored [Cout][k][k][Cin];
  * in a train step the reference encodes the *current* frame twice
    (dynamics.py:584 and :619 -- the posterior never sees the next frame).  Both
    passes produce identical activations, so the encoder runs once; BatchNorm
    running stats receive the two momentum updates and the gradient of both uses
    of `h` flows through the single pass.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_NONE

ENC_PLAN = (("c1", (None, 64, 64)), ("c2", (64, 128, 128)), ("c3", (128, 256, 256, 256)),
            ("c4", (256, 512, 512, None)))
DEC_PLAN = (("upc2", (None, 512, 512, 256)), ("upc3", (512, 256, 256, 128)), ("upc4", (256, 128, 64)),
            ("upc5", (128, 64)))


def _cl_weight(cout: int, cin: int, k: int) -> nn.Parameter:
    """(cout, cin, k, k) parameter with [cout][k][k][cin] memory."""
    mem = torch.empty(cout, k, k, cin)
    return nn.Parameter(mem.permute(0, 3, 1, 2))


class _Conv(nn.Module):
    """Parameter holder named like nn.Conv2d / nn.ConvTranspose2d (`weight`, `bias`)."""

    def __init__(self, cin, cout, k, bias=True, transposed=False):
        super().__init__()
        self.ksize = k
        # ConvTranspose2d stores (cin, cout, k, k)
        self.weight = _cl_weight(cin, cout, k) if transposed else _cl_weight(cout, cin, k)
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None


class _BatchNorm(nn.Module):
    """Holder with nn.BatchNorm2d's parameter / buffer names."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c))
        self.bias = nn.Parameter(torch.empty(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long))
        self.pending_updates = 0  # folded into num_batches_tracked lazily (state_dict time)


class _VggLayer(nn.Module):
    """vgg_layer (vgg_64.py:8-18): `main.0` conv (no bias), `main.1` batch norm; LeakyReLU has no state."""

    def __init__(self, cin, cout):
        super().__init__()
        self.main = nn.ModuleList([_Conv(cin, cout, 3, bias=False), _BatchNorm(cout)])
        self._folded = None

    def folded(self):
        """eval: BatchNorm folded into the conv epilogue as (scale, shift) per output channel."""
        if self._folded is None:
            bn = self.main[1]
            with torch.no_grad():
                scale = bn.weight / torch.sqrt(bn.running_var + ops.BN_EPS)
                self._folded = (scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous())
        return self._folded

    def forward(self, x0, x1=None, n_updates=1, groups=1):
        """`groups` > 1: the batch holds that many time steps, each normalised with its own batch statistics."""
        conv, bn = self.main[0], self.main[1]
        if self.training:
            bn.pending_updates += n_updates * groups
            folded = None
        else:
            folded = self.folded()
        return ops.VggLayer.apply(x0, x1, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                  self.training, n_updates, folded, groups if self.training else 1)


class _Encoder(nn.Module):
    """ConvEncoder (vgg_64.py:87-129)."""

    def __init__(self, dim, nc):
        super().__init__()
        for name, chans in ENC_PLAN:
            chans = [nc if (c is None and i == 0) else (dim if c is None else c) for i, c in enumerate(chans)]
            setattr(self, name, nn.ModuleList([_VggLayer(chans[i], chans[i + 1]) for i in range(len(chans) - 1)]))

    def forward(self, x, n_updates=1, groups=1, first_done=False):
        """`first_done`: x is already the output of c1[0] (the frozen model's direct first-layer kernel)."""
        skips = []
        frozen = not self.training and not torch.is_grad_enabled()
        pooled = None
        for i, name in enumerate(("c1", "c2", "c3", "c4")):
            if i:
                x = pooled if pooled is not None else ops.MaxPool2.apply(x)
                pooled = None
            layers = getattr(self, name)
            for j, layer in enumerate(layers):
                if first_done and i == 0 and j == 0:
                    continue
                if frozen and i < 3 and j == len(layers) - 1:
                    # the frozen model: the block's last layer also writes the pooled map the next block reads
                    both = ops.vgg_pool_frozen(x, layer.main[0].weight, *layer.folded())
                    if both is not None:
                        x, pooled = both
                        continue
                x = layer(x, None, n_updates, groups)
            skips.append(x)
        return x, skips


class _Decoder(nn.Module):
    """ConvDecoder (vgg_64.py:196-241); the skip concat is virtual (second conv source)."""

    def __init__(self, dim, nc):
        super().__init__()
        for name, chans in DEC_PLAN[:-1]:
            chans = [dim if c is None else c for c in chans]
            setattr(self, name, nn.ModuleList([_VggLayer(chans[i], chans[i + 1]) for i in range(len(chans) - 1)]))
        self.upc5 = nn.ModuleList([_VggLayer(128, 64), _Conv(64, nc, 3, bias=True, transposed=True)])

    def forward(self, vec, skip, groups=1):
        d = vec
        for layer in self.upc2:
            d = layer(d, None, 1, groups)
        frozen = not self.training and not torch.is_grad_enabled()

        def up_conv(layer, d, sk):
            """layer(cat[UpsamplingNearest2d(2)(d), sk]); the frozen model reads d at half resolution instead."""
            if frozen and ops.vgg_up_frozen_ok(d, sk, layer.main[0].weight):
                return ops.vgg_up_frozen(d, sk, layer.main[0].weight, *layer.folded())
            return layer(ops.Upsample2.apply(d), sk, 1, groups)

        for name, sk in (("upc3", skip[2]), ("upc4", skip[1])):
            layers = getattr(self, name)
            d = up_conv(layers[0], d, sk)
            for layer in list(layers)[1:]:
                d = layer(d, None, 1, groups)
        d = up_conv(self.upc5[0], d, skip[0])
        head = self.upc5[1]
        return ops.ConvTHead.apply(d, head.weight, head.bias, frozen)


class _LstmCell(nn.Module):
    """ConvLSTMCell (lstm.py:109-149)."""

    def __init__(self, g, k):
        super().__init__()
        self.gates = _Conv(2 * g, 4 * g, k)

    def forward(self, x, state):
        h, c = ops.LstmCell.apply(x, state[0], state[1], self.gates.weight, self.gates.bias, torch.is_grad_enabled())
        return ops.tag_amax(h, ops.amax_one(h.device)), c  # |h| = |o * tanh(c)| < 1


class _GroupNorm(nn.Module):
    """Holder with nn.GroupNorm's parameter names (default init ones / zeros: init_weights skips it)."""

    def __init__(self, groups, c):
        super().__init__()
        self.groups = groups
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))

    def forward(self, x):
        return ops.GroupNorm.apply(x, self.weight, self.bias, self.groups)


class _NormLstmCell(nn.Module):
    """NormConvLSTMCell (lstm.py:151-198, `--lstm_group_norm True`): separate input / hidden gate convs, each
    followed by GroupNorm(16, 4g); GroupNorm(16, g) on the cell state."""

    def __init__(self, g, k):
        super().__init__()
        self.ih_gates = nn.ModuleList([_Conv(g, 4 * g, k), _GroupNorm(16, 4 * g)])
        self.hh_gates = nn.ModuleList([_Conv(g, 4 * g, k), _GroupNorm(16, 4 * g)])
        self.c_norm = _GroupNorm(16, g)

    def forward(self, x, state):
        h_prev, c_prev = state
        ci, ch = self.ih_gates[0], self.hh_gates[0]
        frozen = not torch.is_grad_enabled()
        if frozen and x.is_cuda and ops.norm_cell_frozen_ok(x.shape[3]):
            # no tape: everything behind the two gate convs is one launch (rac_norm_lstm_cell_fwd)
            n_ih, n_hh, n_c = self.ih_gates[1], self.hh_gates[1], self.c_norm
            if ops.is_zero(h_prev):  # a rollout's first step: the conv of an all-zero map is its bias, to the bit
                g_hh = ch.bias.detach().view(1, 1, 1, -1).expand(tuple(x.shape[:3]) + (ch.bias.numel(),)).contiguous()
            else:
                g_hh = ops.ConvBias.apply(h_prev, None, ch.weight, ch.bias, ACT_NONE, True)
            h, c = ops.norm_cell_frozen(ops.ConvBias.apply(x, None, ci.weight, ci.bias, ACT_NONE, True), g_hh, c_prev,
                                        (n_ih.weight, n_ih.bias), (n_hh.weight, n_hh.bias), (n_c.weight, n_c.bias))
            return ops.tag_amax(h, ops.amax_one(h.device, h.shape[0])), c  # |h| = |o * tanh(c)| < 1
        if not frozen and ops.norm_cell_node_ok(x, ci.weight, ch.weight):
            # training: the whole cell is one autograd node (same kernels, none of the bookkeeping between seven nodes)
            n_ih, n_hh, n_c = self.ih_gates[1], self.hh_gates[1], self.c_norm
            h, c = ops.NormLstmCell.apply(x, h_prev, c_prev, ci.weight, ci.bias, n_ih.weight, n_ih.bias, ch.weight, ch.bias,
                                          n_hh.weight, n_hh.bias, n_c.weight, n_c.bias)
            return ops.tag_amax(h, ops.amax_one(h.device)), c  # |h| = |o * tanh(c)| < 1
        g_ih = self.ih_gates[1](ops.ConvBias.apply(x, None, ci.weight, ci.bias, ACT_NONE, frozen))
        g_hh = self.hh_gates[1](ops.ConvBias.apply(h_prev, None, ch.weight, ch.bias, ACT_NONE, frozen))
        c_raw, act = ops.NormCellCore.apply(g_ih, g_hh, c_prev)
        c = self.c_norm(c_raw)
        h = ops.LstmOut.apply(act, c)
        return ops.tag_amax(h, ops.amax_one(h.device)), c  # |h| = |o * tanh(c)| < 1


class _ConvLSTM(nn.Module):
    """ConvLSTM (lstm.py:201-257): layer 0 is 5x5, layer 1 is 3x3; state lives in `self.hidden`."""

    def __init__(self, config, g):
        super().__init__()
        Cell = _NormLstmCell if getattr(config, "lstm_group_norm", False) else _LstmCell
        self.hid_ch = g
        self.lstm = nn.ModuleList([Cell(g, 5), Cell(g, 3)])
        self.batch_size = config.batch_size
        self._hw = (config.image_height // 8, config.image_width // 8)
        self.hidden = None
        self._zero_state = None

    def init_hidden(self, batch_size=None):
        b = self.batch_size if batch_size is None else batch_size
        dev = next(self.parameters()).device
        h, w = self._hw
        key = (b, h, w, self.hid_ch, dev)
        if self._zero_state is None or self._zero_state[0] != key:
            # the initial state is only ever read: one zero map serves every layer, h and c, and every later call
            z0 = ops.tag_amax(torch.zeros(b, h, w, self.hid_ch, device=dev), ops.amax_one(dev))
            z0._rac_zero = True  # the first step's gate conv skips the hidden half of K (ops.is_zero)
            self._zero_state = (key, z0)
        z0 = self._zero_state[1]
        return [(z0, z0) for _ in self.lstm]

    def forward(self, x):
        for i, cell in enumerate(self.lstm):
            self.hidden[i] = cell(x, self.hidden[i])
            x = self.hidden[i][0]
        return x


class _GaussianConvLSTM(_ConvLSTM):
    """GaussianConvLSTM (lstm.py:260-286)."""

    def __init__(self, config, g, z):
        super().__init__(config, g)
        self.mu_net = _Conv(g, z, 3)
        self.logvar_net = _Conv(g, z, 3)
        self._head = None  # (weight, bias) views over both heads, adjacent in the flat buffer (SVGConvModel._flatten)

    def forward(self, x, eps_fn, need_z=True, defer_head=False):
        """`defer_head`: only step the ConvLSTM and draw (and drop) this step's noise in the reference's order; returns
        (None, h, None) -- the caller applies `heads` to the hidden states of all time steps at once."""
        h = super().forward(x)
        if defer_head:
            if need_z:  # the draw z = eps * sigma + mu would have consumed: a (B, H, W, z) map, whatever g is
                eps_fn(h.new_empty(tuple(h.shape[:3]) + (self.mu_net.weight.shape[0],)))
            return None, h, None
        mu, logvar = self.heads(h)
        z = ops.Reparam.apply(mu, logvar, eps_fn(mu)) if need_z else None
        return z, mu, logvar

    def heads(self, h):
        """mu, logvar = mu_net(h), logvar_net(h) (lstm.py:273-274)."""
        head = self._head
        if head is not None and ops.gauss_head_ok(h.shape, head[0]):
            for t in head:  # the views follow the parameters they alias
                if t.requires_grad != self.mu_net.weight.requires_grad:
                    t.requires_grad_(self.mu_net.weight.requires_grad)
            mu, logvar = ops.GaussHead.apply(h, head[0], head[1], not torch.is_grad_enabled())
        else:
            frozen = not torch.is_grad_enabled()
            mu = ops.ConvBias.apply(h, None, self.mu_net.weight, self.mu_net.bias, ACT_NONE, frozen)
            logvar = ops.ConvBias.apply(h, None, self.logvar_net.weight, self.logvar_net.bias, ACT_NONE, frozen)
        return mu, logvar


class SVGConvModel(nn.Module):
    """Conv SVG LSTM predictor (reference dynamics.py:457-644) on hand-written gfx950 kernels."""

    def __init__(self, config):
        super().__init__()
        self._config = cf = config
        self._device = config.device
        self._image_width = cf.image_width
        self._image_height = cf.image_height
        self.eps_source = None  # optional callable(shape_bzhw) -> N(0,1) tensor; tests inject the reference's draws
        self.sequence_batched = None  # forward_sequence_maps: (mu, logvar, mu_p, logvar_p) over all T*B samples, if batched
        self.used_recurrent_core = False  # did the last forward_sequence_maps take ops.RecurrentCore?
        self._flat = self._flat_grad = None
        if cf.image_width not in (64, 128):  # dynamics.py:470-473
            raise ValueError
        enc_c = cf.channels
        if cf.model_use_mask:
            enc_c += 1 + (1 if cf.model_use_future_mask else 0)
        if getattr(cf, "model_use_heatmap", False):
            enc_c += 1 + (1 if getattr(cf, "model_use_future_heatmap", False) else 0)
        g, z, A, R = cf.g_dim, cf.z_dim, cf.action_dim, cf.robot_dim
        use_r, use_rn = cf.model_use_robot_state, cf.model_use_future_robot_state
        extra = (R if use_r else 0) + (R if use_rn else 0)
        self.encoder = _Encoder(g, enc_c)
        self.frame_pred_input_conv = _Conv(g + A + z + extra, g, 3)
        self.frame_predictor = _ConvLSTM(cf, g)
        self.posterior_input_conv = _Conv(g + (R if use_r else 0), g, 3)
        self.prior_input_conv = _Conv(g + A + extra, g, 3)
        self.posterior = _GaussianConvLSTM(cf, g, z)
        self.prior = _GaussianConvLSTM(cf, g, z)
        self.decoder = _Decoder(g, cf.channels + 1)
        self.reset_parameters()
        self.to(self._device)

    # ------------------------------------------------------------------ params
    def reset_parameters(self):
        """init_weights (base.py:26-36): conv W ~ N(0, 0.02), b = 0; BatchNorm gamma ~ N(1, 0.02), beta = 0."""
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, _Conv):
                    m.weight.normal_(0.0, 0.02)
                    if m.bias is not None:
                        m.bias.zero_()
                elif isinstance(m, _BatchNorm):
                    m.weight.normal_(1.0, 0.02)
                    m.bias.zero_()

    def _flatten(self):
        """Re-home every parameter as a view of one flat buffer (and .grad of another)."""
        params = list(self.parameters())
        if not params:
            return
        dev = params[0].device
        # buffer order = registration order, except that the mu / logvar heads of a Gaussian LSTM lie weight next to
        # weight and bias next to bias: stacked along Cout they are ONE conv (ops.GaussHead)
        heads = [m for m in self.modules() if isinstance(m, _GaussianConvLSTM)]
        paired = {id(p) for m in heads for c in (m.mu_net, m.logvar_net) for p in (c.weight, c.bias)}
        order = [p for p in params if id(p) not in paired]
        for m in heads:
            order += [m.mu_net.weight, m.logvar_net.weight, m.mu_net.bias, m.logvar_net.bias]
        total = 0
        for p in order:
            p._rac_off = total
            total += (p.numel() + 3) // 4 * 4  # keep every view 16-byte aligned
        # zero_grad(lazy=True): the large conv weights' gradients are not zeroed (ops._STALE), everything else is -- by ONE
        # multi-tensor fill.  (Moving the large weights to the end of the buffer instead, so that one slice fill would do,
        # slowed the fused Adam pass from 1.7 to 2.8 ms: its blocks then walk the three 210 MB gate weights back to back.)
        total = (total + 1023) // 1024 * 1024  # whole 4 KB: any 1 / 2 / 4 / 8-way slice of a bucket stays 16-byte aligned
        self._lazy_params = [p for p in params if id(p) not in paired and p.dim() == 4 and p.numel() >= (1 << 16)]
        lazy_ids = {id(p) for p in self._lazy_params}
        self._eager_params = [p for p in params if id(p) not in lazy_ids]
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        grad = torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p in params:
                view = torch.as_strided(flat, p.shape, p.stride(), p._rac_off)
                view.copy_(p.data)
                p.data = view
                p.grad = torch.as_strided(grad, p.shape, p.stride(), p._rac_off)
        self._flat, self._flat_grad = flat, grad
        for m in heads:
            w0, w1, b0, b1 = m.mu_net.weight, m.logvar_net.weight, m.mu_net.bias, m.logvar_net.bias
            m._head = None
            if w0._rac_off + w0.numel() == w1._rac_off and b0._rac_off + b0.numel() == b1._rac_off:
                z, g, k, _ = w0.shape
                shape, stride = (2 * z, g, k, k), (k * k * g, 1, k * g, g)
                wm = torch.as_strided(flat, shape, stride, w0._rac_off).requires_grad_(w0.requires_grad)
                wm.grad = torch.as_strided(grad, shape, stride, w0._rac_off)
                bm = torch.as_strided(flat, (2 * z,), (1,), b0._rac_off).requires_grad_(w0.requires_grad)
                bm.grad = torch.as_strided(grad, (2 * z,), (1,), b0._rac_off)
                wm._rac_sources = (w0, w1)  # ops.weight_parts: the cached operand parts follow both parameters
                m._head = (wm, bm)
        for m in self.modules():
            if isinstance(m, _VggLayer):
                m._folded = None

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def flat_parameters(self):
        return self._flat, self._flat_grad

    def zero_grad(self, set_to_none: bool = False, lazy: bool = False):
        """Gradients are accumulated in place by the kernels: zero the flat buffer, keep the views attached.
        `lazy` (PredictionTrainer._train_step): the large conv weights' gradients are only marked stale -- their
        weight-gradient launch overwrites them, `ops.finish_grads()` zeroes what nothing wrote (ops._STALE)."""
        if self._flat_grad is None:
            self._flatten()
        ops._STALE.clear()
        if not (lazy and self._lazy_params):
            # an optimiser update still running on a side stream (optim.FusedAdam.overlap_next_forward) READS the late
            # weights' gradients: a whole-buffer fill on this stream must come behind it (the lazy form never touches them)
            ops.param_wait()
            self._flat_grad.zero_()
        for p in self.parameters():
            off = p._rac_off
            if p.grad is None or p.grad.data_ptr() != self._flat_grad.data_ptr() + 4 * off:
                p.grad = torch.as_strided(self._flat_grad, p.shape, p.stride(), off)
        if lazy and self._lazy_params:
            torch._foreach_zero_([p.grad for p in self._eager_params])
            for p in self._lazy_params:
                ops.mark_stale(p.grad)

    def train(self, mode: bool = True):
        for m in self.modules():
            if isinstance(m, _VggLayer):
                m._folded = None
        return super().train(mode)

    def _flush_bn_counters(self):
        for m in self.modules():
            if isinstance(m, _BatchNorm) and m.pending_updates:
                m.num_batches_tracked += m.pending_updates
                m.pending_updates = 0

    def state_dict(self, *a, **k):
        # (an optimiser update still in flight -- FusedAdam's late weights on a side stream, ShardedAdam's all-gather --
        # must land before anyone reads the parameters through the returned views)
        ops.param_wait()
        self._flush_bn_counters()
        sd = super().state_dict(*a, **k)
        return sd

    def load_state_dict(self, state_dict, strict: bool = True, **k):
        out = super().load_state_dict(state_dict, strict=strict, **k)
        for m in self.modules():
            if isinstance(m, _VggLayer):
                m._folded = None
            if isinstance(m, _BatchNorm):
                m.pending_updates = 0
        return out

    # ------------------------------------------------------------------ state
    def init_hidden(self, batch_size=None):
        """Initialize the recurrent states by batch size (dynamics.py:536-542)."""
        self.frame_predictor.hidden = self.frame_predictor.init_hidden(batch_size)
        self.posterior.hidden = self.posterior.init_hidden(batch_size)
        self.prior.hidden = self.prior.init_hidden(batch_size)

    def _eps(self, like_map: torch.Tensor) -> torch.Tensor:
        b, h, w, z = like_map.shape
        if self.eps_source is not None:
            e = self.eps_source((b, z, h, w)).to(like_map.device, torch.float32)
            return e.permute(0, 2, 3, 1).contiguous()
        return torch.randn(b, h, w, z, device=like_map.device)

    # ---------------------------------------------------------------- forward
    def forward(self, image, mask, robot, heatmap, action, next_image=None, next_mask=None, next_robot=None,
                next_heatmap=None, skip=None, force_use_prior=False, sample_mean=False):
        """Predict the next frame (same contract as reference dynamics.py:544-644).

        Returns (x_pred (B,4,H,W) in [0,1], skip list, mu, logvar, mu_p, logvar_p); mu/logvar are None
        when `next_image` is None."""
        x4, skip_maps, mu, logvar, mu_p, logvar_p = self.forward_maps(
            image, mask, robot, heatmap, action, next_image is not None, next_robot,
            None if skip is None else [s.permute(0, 2, 3, 1).contiguous() for s in skip],
            force_use_prior, sample_mean)
        v = ops.to_planes_view
        return (v(x4), [v(s) for s in skip_maps], None if mu is None else v(mu),
                None if logvar is None else v(logvar), v(mu_p), v(logvar_p))

    def forward_maps(self, image, mask, robot, heatmap, action, posterior: bool, next_robot=None, skip=None,
                     force_use_prior=False, sample_mean=False, zero_mask=None, shared_frame=False):
        """`forward` on NHWC maps (no layout conversion of the results).  `zero_mask` fuses
        zero_robot_region(mask, image) (src/utils/image.py:5-19) into the input packing.
        `shared_frame` (frozen model): every sample's frame / mask / heatmap is the same one (a planner's first step,
        trajectory_sampler.py:131-137): the encoder runs on one image and its maps are copied to the batch -- the same
        bits, since every image is scaled by its own maximum."""
        if shared_frame and not self.training and not torch.is_grad_enabled() and image.shape[0] > 1:
            n = image.shape[0]
            one = lambda t: None if t is None else t[:1]
            h, curr_skip = self._encode(one(image), one(mask), one(heatmap), one(zero_mask), 1, 1)
            h = ops.broadcast_images(h, n)
            curr_skip = [ops.broadcast_images(s, n) for s in curr_skip]
        else:
            h, curr_skip = self._encode(image, mask, heatmap, zero_mask, 2 if posterior else 1, 1)
        if self._config.last_frame_skip or skip is None:
            skip = curr_skip
        h_pred, mu, logvar, mu_p, logvar_p = self._recur(h, robot, action, posterior, next_robot, force_use_prior,
                                                         sample_mean)
        x4 = self.decoder(h_pred, skip)
        return x4, skip, mu, logvar, mu_p, logvar_p

    def forward_sequence_maps(self, images, masks, robots, heatmaps, actions, next_robots, zero_masks=None):
        """Teacher-forced training window in one pass (all `T` inputs are ground truth, `last_frame_skip`):
        the encoder runs ONCE over the T*B input frames and the decoder ONCE over the T*B predicted latents, every
        BatchNorm layer keeping one set of batch statistics per time step (and applying their running-stat updates
        in time order), so each sample sees exactly the arithmetic of T calls of `forward_maps`; only the
        recurrent part (input convs, prior / posterior / frame-predictor ConvLSTMs) is stepped.

        images (T,B,3,H,W), masks (T,B,m,H,W) or None, heatmaps likewise, zero_masks (T,B,1,H,W) or None;
        robots / actions / next_robots: per-step sequences (robots[t] may be a (r, r_next) tuple).
        Returns (x4 (T*B,H,W,4), [mu_t], [logvar_t], [mu_p_t], [logvar_p_t])."""
        T, B = images.shape[0], images.shape[1]
        flat = lambda t: None if t is None else t.reshape((T * B,) + tuple(t.shape[2:]))
        h_all, skips = self._encode(flat(images), flat(masks), flat(heatmaps), flat(zero_masks), 2, T, staged=True)
        h_steps = h_all.view((T, B) + tuple(h_all.shape[1:])).unbind(0)
        # the prior's and the posterior's input convs see only data and the encoder output (the posterior encodes the
        # CURRENT frame: dynamics.py:619), all known before the recurrence starts: ONE launch each over the T*B latents
        # instead of T (M = 5120 instead of 5 x 1024 at cfg2: no K split, no combine, a fifth of the launches, forward
        # and backward); only the frame predictor's input conv waits for z_t
        cf = self._config
        cat = lambda seq: torch.cat([v.contiguous() for v in seq], 0)
        a_all = cat(actions)
        r_all = rn_all = None
        if cf.model_use_robot_state:
            if cf.model_use_future_robot_state:
                r_all, rn_all = cat([r[0] for r in robots]), cat([r[1] for r in robots])
            else:
                r_all = cat(robots)
        prior_all = self._embed(self.prior_input_conv, [v for v in (a_all, r_all, rn_all) if v is not None], h_all, None)
        if cf.model_use_robot_state:
            post_all = self._embed(self.posterior_input_conv, [cat(next_robots)], h_all, None)
        else:
            q = self.posterior_input_conv
            post_all = ops.ConvBias.apply(h_all, None, q.weight, q.bias, ACT_NONE, not torch.is_grad_enabled())
        core = self._recurrent_core(T, B, h_all, prior_all, post_all, robots, actions)
        ops.param_wait()  # (whatever the core did not wait for itself: every path from here on may read any parameter)
        if core is not None:
            h_pred_all, mu_all, lv_all, h_prior_all = core
            mu_p_all, logvar_p_all = self.prior.heads(h_prior_all)
            x4 = self.decoder(h_pred_all, skips, T)
            # (the trainer's KL term takes the batched tensors: one launch, no per-step slices for autograd to stack)
            self.sequence_batched = (mu_all, lv_all, mu_p_all, logvar_p_all)
            self.used_recurrent_core = True
            per_step = lambda t_: list(t_.view((T, B) + tuple(t_.shape[1:])).unbind(0))
            return x4, per_step(mu_all), per_step(lv_all), per_step(mu_p_all), per_step(logvar_p_all)
        self.sequence_batched = None
        self.used_recurrent_core = False
        # (a step's slice inherits the whole tensor's maximum: a valid bound, and no reduction pass per step)
        steps_of = lambda t_: [ops.retag(s_, ops.amax_tag(t_)) for s_ in t_.view((T, B) + tuple(t_.shape[1:])).unbind(0)]
        prior_steps, post_steps = steps_of(prior_all), steps_of(post_all)
        h_preds, mus, logvars, mu_ps, logvar_ps = [], [], [], [], []
        for t in range(T):
            # the prior's z is never used on this path (the posterior's drives the frame predictor): its mu / logvar heads
            # only feed the KL term, so they too run once over all steps' hidden states, after the loop
            h_pred, mu, logvar, h_prior, _ = self._recur(h_steps[t], robots[t], actions[t], True, next_robots[t],
                                                         False, False, prior_in=prior_steps[t], post_in=post_steps[t],
                                                         defer_prior_head=True)
            h_preds.append(h_pred)
            mus.append(mu)
            logvars.append(logvar)
            mu_ps.append(h_prior)
        mu_p_all, logvar_p_all = self.prior.heads(torch.cat(mu_ps, 0))
        mu_ps = list(mu_p_all.view((T, B) + tuple(mu_p_all.shape[1:])).unbind(0))
        logvar_ps = list(logvar_p_all.view((T, B) + tuple(logvar_p_all.shape[1:])).unbind(0))
        x4 = self.decoder(torch.cat(h_preds, 0), skips, T)
        return x4, mus, logvars, mu_ps, logvar_ps

    def _recurrent_core(self, T, B, h_all, prior_all, post_all, robots, actions):
        """The T-step recurrence as ONE hand-scheduled autograd node (ops.RecurrentCore; ops.NormRecurrentCore for
        `--lstm_group_norm True`) where its kernels apply; None otherwise (narrow models, frozen parameters: the per-step
        autograd path below)."""
        cf = self._config
        lstms = {"prior": self.prior, "post": self.posterior, "fp": self.frame_predictor}
        all_cells = [c for m in lstms.values() for c in m.lstm]
        norm = all(isinstance(c, _NormLstmCell) for c in all_cells)  # --lstm_group_norm True: ops.NormRecurrentCore
        if not norm and any(not isinstance(c, _LstmCell) for c in all_cells):
            return None
        g, z = cf.g_dim, cf.z_dim
        vs = []
        for t in range(T):
            a = actions[t].contiguous()
            if cf.model_use_robot_state:
                r = robots[t]
                vs.append([a] + ([r[0].contiguous(), r[1].contiguous()] if cf.model_use_future_robot_state
                                 else [r.contiguous()]))
            else:
                vs.append([a])
        nv = sum(v.shape[1] for v in vs[0])
        head = self.posterior._head
        cells = all_cells
        ok = ops.norm_recurrent_core_ok if norm else ops.recurrent_core_ok
        if not ok(h_all, g, z, nv, cells, head, self.frame_pred_input_conv):
            return None
        for t in head:  # the merged views follow the parameters they alias
            if t.requires_grad != self.posterior.mu_net.weight.requires_grad:
                t.requires_grad_(self.posterior.mu_net.weight.requires_grad)
        plan = dict(T=T, B=B, g=g, z=z, nv=nv, vs=vs, cells={k: tuple(m.lstm) for k, m in lstms.items()}, head=head,
                    frame_conv=self.frame_pred_input_conv, init_state={k: m.hidden for k, m in lstms.items()},
                    eps_fn=self._eps, draw_prior_noise=True)
        if norm:
            params = ([c.ih_gates[0].weight for c in cells] + [c.hh_gates[0].weight for c in cells]
                      + [head[0], self.frame_pred_input_conv.weight])
            out = ops.NormRecurrentCore.apply(plan, h_all, prior_all, post_all, *params)
        else:
            params = [c.gates.weight for c in cells] + [head[0], self.frame_pred_input_conv.weight]
            out = ops.RecurrentCore.apply(plan, h_all, prior_all, post_all, *params)
        for k, m in lstms.items():
            m.hidden = plan["final_state"][k]
        return out

    def sequence_ok(self, batch: int, height: int, width: int) -> bool:
        """`forward_sequence_maps` needs per-step row ranges that are whole 128-row tiles at every resolution."""
        return self.training and bool(self._config.last_frame_skip) and (batch * (height // 8) * (width // 8)) % 128 == 0

    def _encoder_extent(self) -> int:
        """Flat-buffer element index behind the encoder's last parameter (the encoder is registered first)."""
        ext = getattr(self, "_enc_extent", None)
        if ext is None or ext[0] is not self._flat:  # (cached per flat buffer: walking the modules costs 50 us per call)
            ext = self._enc_extent = (self._flat, max(p._rac_off + p.numel() for p in self.encoder.parameters()))
        return ext[1]

    def late_update_groups(self):
        """Parameters behind the encoder's in the order a teacher-forced window first reads them (optim.FusedAdam updates
        its late weights in these groups): the prior's chain (+ both input convs, which run right behind the encoder), the
        posterior's, then the frame predictor's and the decoder's (everything else)."""
        first = [self.prior_input_conv.weight, self.posterior_input_conv.weight] + [c.gates.weight for c in self.prior.lstm
                                                                                    if hasattr(c, "gates")]
        second = [c.gates.weight for c in self.posterior.lstm if hasattr(c, "gates")]
        return [first, second]

    def _encode(self, image, mask, heatmap, zero_mask, n_updates, groups, staged=False):
        """`staged`: the caller waits for the later parameter groups itself (forward_sequence_maps: the recurrent core takes
        its chains' weights as they arrive); otherwise everything is waited for behind the encoder."""
        gate = ops.PARAM_GATE  # an optimiser update still in flight: optim.ShardedAdam's all-gather, FusedAdam's late groups
        if gate is not None and getattr(gate, "_model", self) is not self:
            gate.wait_params()  # another model's optimiser (a previous trainer of this process): let it land and drop it
            gate = ops.PARAM_GATE = None
        if gate is not None:
            gate.wait_params(upto=self._encoder_extent())
        out = self._encode_now(image, mask, heatmap, zero_mask, n_updates, groups)
        if gate is not None:
            if staged and hasattr(gate, "wait_for"):
                gate.wait_for(self.prior_input_conv.weight)  # the first late group, under the encoder's kernels
            else:
                gate.wait_params()  # everything behind the encoder's parameters: waited for under the encoder's kernels
        return out

    def _encode_now(self, image, mask, heatmap, zero_mask, n_updates, groups):
        cf = self._config
        image = image.contiguous()
        mask_planes = None
        if getattr(cf, "model_use_heatmap", False):
            mask_planes = heatmap
        if cf.model_use_mask:
            mask_planes = mask if mask_planes is None else torch.cat([mask_planes, mask], 1)
        if mask_planes is not None:
            mask_planes = mask_planes.contiguous()
        zm = None if zero_mask is None else zero_mask.contiguous()
        first = self.encoder.c1[0]
        if (not self.training and not torch.is_grad_enabled() and ops.SPLIT_GEMM
                and ops.first_layer_ok(image, mask_planes, first.main[0].weight)):
            # frozen model: the first layer reads the planes directly (no packed, 32-channel-padded input tensor)
            scale, shift = first.folded()
            x1 = ops.first_layer_frozen(image, zm, mask_planes, first.main[0].weight, scale, shift)
            return self.encoder(x1, n_updates, groups, first_done=True)
        if (self.training and torch.is_grad_enabled() and ops.first_layer_train_ok(image, mask_planes, first.main[0].weight)
                and (image.shape[0] * image.shape[-2] * image.shape[-1]) % groups == 0):
            # training, frames are data: the first layer reads the planes on the matrix pipe (ops.FirstVggLayer)
            bn = first.main[1]
            bn.pending_updates += n_updates * groups
            x1 = ops.FirstVggLayer.apply(image, zm, mask_planes, first.main[0].weight, bn.weight, bn.bias, bn.running_mean,
                                         bn.running_var, n_updates, groups)
            return self.encoder(x1, n_updates, groups, first_done=True)
        # one whole 32-channel chunk (zero padded) where the first layer can take the split-precision kernels
        H, W = image.shape[-2], image.shape[-1]
        pad_to = 32 if (ops.SPLIT_GEMM and ops.split_supported(H, W, 3, 32, 64)) else 0
        x_in = ops.PackInput.apply(image, None if zero_mask is None else zero_mask.contiguous(), mask_planes, pad_to)
        return self.encoder(x_in, n_updates, groups)

    def _embed(self, conv, vs, h, z):
        """conv(cat[tile(vs), h, z]) (dynamics.py:591-607,634-640); the frozen model's conv reads h in place (no
        concatenated tensor)."""
        frozen = not torch.is_grad_enabled()  # no tape: the input convs may take the split-precision pipe
        if frozen and not self.training and ops.embed_frozen_ok(vs, h, z, conv.weight):
            return ops.embed_frozen(vs, h, z, conv.weight, conv.bias)
        vs3 = list(vs) + [None] * (3 - len(vs))
        return ops.ConvBias.apply(ops.TileCat.apply(vs3[0], vs3[1], vs3[2], h, z, frozen), None, conv.weight,
                                  conv.bias, ACT_NONE, frozen)

    def _recur(self, h, robot, action, posterior, next_robot, force_use_prior, sample_mean, prior_in=None, post_in=None,
               defer_prior_head=False):
        """The stepped part of `forward`: prior / posterior / frame predictor on one time step's latent.
        `prior_in` / `post_in`: the two input convs' outputs when the caller ran them for all time steps at once;
        `defer_prior_head` (posterior given, its z used): mu_p comes back as the prior's hidden state, logvar_p as None."""
        cf = self._config
        a = action.contiguous()
        r = r_next = None
        if cf.model_use_robot_state:
            if cf.model_use_future_robot_state:
                r, r_next = robot
                r, r_next = r.contiguous(), r_next.contiguous()
            else:
                r = robot.contiguous()
        frozen = not torch.is_grad_enabled()
        if prior_in is None:
            prior_in = self._embed(self.prior_input_conv, [v for v in (a, r, r_next) if v is not None], h, None)
        defer = defer_prior_head and posterior and not force_use_prior
        z_p, mu_p, logvar_p = self.prior(prior_in, self._eps, need_z=not sample_mean, defer_head=defer)
        z = mu_p if sample_mean else z_p
        mu = logvar = None
        if posterior:
            q = self.posterior_input_conv
            if post_in is None:
                if cf.model_use_robot_state:
                    post_in = self._embed(q, [next_robot.contiguous()], h, None)
                else:
                    post_in = ops.ConvBias.apply(h, None, q.weight, q.bias, ACT_NONE, frozen)
            z_t, mu, logvar = self.posterior(post_in, self._eps)
            if not force_use_prior:
                z = z_t
        f = self.frame_pred_input_conv
        frame_in = self._embed(f, [v for v in (a, r, r_next) if v is not None], h, z)
        h_pred = self.frame_predictor(frame_in)
        return h_pred, mu, logvar, mu_p, logvar_p
