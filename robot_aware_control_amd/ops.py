"""Host-side operators over the C ABI: thin launch wrappers plus the
`torch.autograd.Function`s that give the SVG model its backward pass.

PyTorch is used for device memory (`torch.empty`), streams and the autograd tape;
every arithmetic op on the hot path is a HIP kernel of librac_hip.so.

Layouts: feature maps are contiguous (B, H, W, C) tensors ("maps", NHWC);
frames at the model boundary are contiguous (B, C, H, W) tensors ("planes");
conv weights are logical (Cout, Cin, k, k) parameters whose MEMORY is
[Cout][k][k][Cin] (channels_last strides).

Weight / bias / BatchNorm-affine gradients are accumulated by the kernels
straight into `param.grad` (the model keeps one flat gradient buffer, zeroed by
`zero_grad()`), so the corresponding autograd outputs are None: BPTT over the
time steps then costs no extra read-modify-write passes over the 954 MB of
weights.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import functools
import os
import weakref
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import AbsmaxJob, AdamFragJob, AdamRange, ConvArgs, FragJob, GradSrc, WgradArgs, call, ptr, stream_ptr

FWD, DGRAD, WGRAD = 0, 1, 2
ACT_NONE, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2
LOSS_KINDS = {"mse": 0, "l1": 1, "dontcare_mse": 2, "dontcare_l1": 3}
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _require_cuda(t: torch.Tensor):
    if not t.is_cuda:
        raise _lib.RacError("robot_aware_control_amd ops run on the GPU only (no CPU fallback); got a CPU tensor")


def weight_mem(w: torch.Tensor) -> torch.Tensor:
    """Check that a (Cout,Cin,k,k) weight is stored [Cout][k][k][Cin]."""
    co, ci, kh, kw = w.shape
    if w.stride() != (kh * kw * ci, 1, kw * ci, ci):
        raise _lib.RacError(f"conv weight {tuple(w.shape)} must use channels_last memory, strides {w.stride()}")
    return w


# Lazy zero_grad (the trainer's step only): the large conv weights' gradients are NOT zeroed at the start of a step --
# their one time-batched weight-gradient launch WRITES them (rac_wgrad_args.accumulate = 0) instead of adding to zeros,
# which saves the 954 MB fill and the read half of the read-modify-write.  A buffer listed here is "stale": whoever
# touches it first either overwrites it whole (conv_wgrad_split_acc) or zeroes it on the spot (grad_buffer);
# finish_grads() zeroes what nobody wrote, before anything reads gradients.
_STALE = {}
LAZY_ZERO_GRAD = os.environ.get("RAC_LAZY_ZERO_GRAD", "1") == "1"


def mark_stale(g: torch.Tensor) -> None:
    _STALE[g.data_ptr()] = g


def take_stale(g: torch.Tensor) -> bool:
    return bool(_STALE) and _STALE.pop(g.data_ptr(), None) is not None


def finish_grads() -> None:
    for g in _STALE.values():
        g.zero_()
    _STALE.clear()


def grad_buffer(p: torch.Tensor) -> torch.Tensor:
    """`p.grad`, allocated (zeroed, same strides) on first use -- and zeroed now if a lazy zero_grad left it stale."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)  # preserve_format keeps the channels_last strides
    elif _STALE and take_stale(p.grad):
        p.grad.zero_()
    return p.grad


def _cdiv(a, b):
    return (a + b - 1) // b


# Convs whose input-channel count is not a multiple of 4 (first encoder layer: 3..7 channels; the three input
# convs: g + 5..15 (+ z)) run on a zero-padded copy of the weight ([Cout][k][k][Cpad], 16-byte rows) so that they
# take the vector-load kernel; their inputs are produced already padded (rac_pack_input / rac_tilecat_fwd `pad`).
PARAM_EPOCH = 0  # bumped by FusedAdam.step(): parameters changed behind torch's version counters


def pad4(c: int) -> int:
    return (-c) % 4


def _derived(weight: torch.Tensor, slot: str, build):
    """Cache a tensor derived from a parameter ON the parameter object (dies with it; a data_ptr key could be
    re-used by the allocator), invalidated by in-place updates (torch version counter or PARAM_EPOCH)."""
    tag = (weight._version, PARAM_EPOCH, weight.data_ptr())
    hit = getattr(weight, slot, None)
    if hit is not None and hit[0] == tag:
        return hit[1]
    val = build()
    setattr(weight, slot, (tag, val))
    return val


def padded_weight(weight: torch.Tensor, cp: Optional[int] = None) -> torch.Tensor:
    """Zero-padded (Cout, cp, k, k) channels_last copy of `weight` (cp defaults to Cin rounded up to a multiple of 4),
    cached until the parameter changes."""
    co, ci, k, _ = weight.shape
    cp = ci + pad4(ci) if cp is None else cp

    def build():
        # the copy keeps its buffer (and its identity: the split-precision parts registry keys on the object)
        wp = getattr(weight, f"_rac_padbuf{cp}", None)
        if wp is None or wp.device != weight.device:
            wp = torch.empty((co, k, k, cp), device=weight.device, dtype=torch.float32).permute(0, 3, 1, 2)
            setattr(weight, f"_rac_padbuf{cp}", wp)
            wp._rac_pad_source = (weakref.ref(weight), cp)
        call("rac_pad_rows", ptr(weight_mem(weight)), ci, ptr(wp), cp, co * k * k, stream_ptr())
        ent = _WP_ENTRIES.get(id(wp))
        if ent is not None:  # rewritten behind torch's version counter: its cached operand parts are stale
            ent.tag = None
        return wp
    return _derived(weight, f"_rac_padded{cp}", build)


def wgrad_padded_acc(dy, x0, weight):
    """weight.grad += unpad(dW_pad) for a conv that ran on a padded weight / padded input x0."""
    co, ci, k, _ = weight.shape
    cp = x0.shape[3]
    gp = torch.zeros((co, k, k, cp), device=dy.device, dtype=torch.float32).permute(0, 3, 1, 2)
    B, H, W, Cout = dy.shape
    conv_raw(WGRAD, x0, None, dy, gp, B=B, H=H, W=W, ksize=k, Cin=cp, Cout=Cout, a_split=cp, accumulate=1, split_k=0)
    call("rac_unpad_add", ptr(gp), cp, ptr(weight_mem(grad_buffer(weight))), ci, co * k * k, stream_ptr())


def thin_wgrad_acc(wide, thin, ct: int, weight):
    """weight.grad ([64][3][3][ct] memory) += the 3x3 conv weight gradient between a 64-channel map and the first `ct`
    channels of a thin one (rac_thin_wgrad: first encoder layer, output head): partial sums per workgroup, added in a
    fixed order."""
    B, H, W, _ = wide.shape
    n = 64 * 9 * ct
    n_parts = min(512, B * (H // 16) * (W // 16))
    parts = torch.empty((n_parts, n), device=wide.device, dtype=torch.float32)
    sp = stream_ptr()
    call("rac_thin_wgrad", ptr(wide), ptr(thin), thin.shape[3], ct, ptr(parts), n_parts, B, H, W, 64, sp)
    call("rac_slab_accumulate", ptr(parts), n_parts, n, ptr(weight_mem(grad_buffer(weight))), n, sp)


MAX_SPLIT_K = int(os.environ.get("RAC_MAX_SPLIT_K", "8"))  # most K splits of a split-precision forward / data-gradient launch


def plan_split_k(M: int, N: int, nchunks: int, tile128_only: bool = False) -> int:
    """K-splits of a FWD/DGRAD launch so that >= ~2 workgroups land on each of the 256 CUs.  Mirrors the tile
    choice of rac_conv2d (128x128 when tiles*split >= 192, else 64x64, 128x32 for narrow N)."""
    return _plan_split_k(M, N, nchunks, tile128_only, os.environ.get("RAC_SPLIT"), MAX_SPLIT_K)


@functools.lru_cache(maxsize=None)
def _plan_split_k(M, N, nchunks, tile128_only, forced, max_split):
    if forced:
        return max(1, min(int(forced), nchunks))
    cap = max(1, nchunks // 8)  # keep >= 8 chunks per split: the pipeline prologue/epilogue must amortise
    if tile128_only:  # the split-precision kernel has one tile shape
        t128 = _cdiv(M, 128) * _cdiv(N, 128)
        if t128 >= 384:
            return 1
        # 512 workgroups run at a time (2 per CU): the split whose LAST round is fullest, a little cheaper the smaller it
        # is (each split is one more slab to write and combine).  Measured on the M = 5120 / 20 480 vgg shapes: 160 tiles
        # x 3 = 480 workgroups 78 us against 88 (x 4 = 640: a quarter-full second round) and 97 (x 2); 320 tiles x 3 = 960
        # 89 us against 98 (x 1: half the slots idle) and 96 (x 4); the gate GEMMs keep 128 x 4 and 64 x 8 = 512.
        best, best_eff = 1, -1.0
        for s in range(1, min(max_split, cap) + 1):
            rounds = t128 * s / 512.0
            eff = (rounds / -(-rounds // 1) if rounds > 1 else rounds) * (1.0 - 0.03 * (s - 1))
            if eff > best_eff + 1e-9:
                best, best_eff = s, eff
        return best
    if N <= 32:
        tiles = _cdiv(M, 128)
        return 1 if tiles >= 384 else max(1, min(8, _cdiv(512, tiles), cap))
    if M >= 128 and N >= 128:
        t128 = _cdiv(M, 128) * _cdiv(N, 128)
        if t128 >= 384:
            return 1
        split = max(1, min(8, _cdiv(512, t128), cap))
        if t128 * split >= 192:
            return split
    tiles = _cdiv(M, 64) * _cdiv(N, 64)
    if tiles >= 384:
        return 1
    return max(1, min(8, _cdiv(512, tiles), cap))


# --------------------------------------------------------------------------- #
# raw launches
# --------------------------------------------------------------------------- #
# bench.py sets PROFILE = {"match": (mode, ksize, Cin, Cout), "events": []} to time ONE kernel shape with
# HIP events on the launch stream (torch.cuda.Event records on the current stream = the launch stream).
PROFILE = None
# RAC_SHAPE_LOG=<file>: every conv launch appends (kernel family, mode, k, M, N, K, algorithmic FLOP, pipe peak) here and
# the list is written to <file> at exit; tools/shape_profile.py joins it with a rocprofv3 kernel trace (same launch
# order) into the per-shape roofline table kept under profiles/.
SHAPE_LOG = [] if os.environ.get("RAC_SHAPE_LOG") else None
if SHAPE_LOG is not None:
    import atexit
    import json as _json
    atexit.register(lambda: _json.dump(SHAPE_LOG, open(os.environ["RAC_SHAPE_LOG"], "w")))


def _log_shape(family, mode, k, M, N, K, peak):
    SHAPE_LOG.append({"family": family, "mode": ("fwd", "dgrad", "wgrad")[mode], "k": k, "M": M, "N": N, "K": K,
                      "flop": 2.0 * M * N * K, "peak": peak})


def conv_raw(mode: int, a0, a1, w, out0, out1=None, *, B, H, W, ksize, Cin, Cout, act=ACT_NONE, split_k=1,
             slab_stride=0, accumulate=0, a_split=0, o_split=0, bias=None, scale=None, shift=None, stats=None,
             stats_rows=0):
    prof = PROFILE
    if prof is not None and prof["match"] == (mode, ksize, Cin, Cout):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _conv_launch(mode, a0, a1, w, out0, out1, B, H, W, ksize, Cin, Cout, act, split_k, slab_stride, accumulate,
                     a_split, o_split, bias, scale, shift, stats, stats_rows)
        e1.record()
        prof["events"].append((e0, e1, B * H * W))
        return
    _conv_launch(mode, a0, a1, w, out0, out1, B, H, W, ksize, Cin, Cout, act, split_k, slab_stride, accumulate,
                 a_split, o_split, bias, scale, shift, stats, stats_rows)


def _conv_launch(mode, a0, a1, w, out0, out1, B, H, W, ksize, Cin, Cout, act, split_k, slab_stride, accumulate,
                 a_split, o_split, bias, scale, shift, stats, stats_rows=0):
    args = ConvArgs(mode=mode, B=B, H=H, W=W, ksize=ksize, Cin=Cin, Cout=Cout, act=act, split_k=split_k,
                    accumulate=accumulate, a_split=a_split, o_split=o_split, slab_stride=slab_stride,
                    a0=ptr(a0), a1=ptr(a1), w=ptr(w), out0=ptr(out0), out1=ptr(out1), bias=ptr(bias),
                    scale=ptr(scale), shift=ptr(shift), stats=ptr(stats), stats_rows=stats_rows)
    if SHAPE_LOG is not None:
        M, N, K = B * H * W, (Cout if mode == FWD else Cin), (Cin if mode == FWD else Cout) * ksize * ksize
        if mode == WGRAD:
            M, N, K = Cout, Cin * ksize * ksize, B * H * W
        _log_shape("igemm", mode, ksize, M, N, K, 157.3)
    call("rac_conv2d", C.byref(args), stream_ptr())


def conv_forward(x0, x1, weight, bias=None, *, act=ACT_NONE, scale=None, shift=None, stats=None,
                 allow_split=True, want_slabs=False, groups=1, frozen=False):
    """FWD conv over the virtual concat [x0 | x1].  Returns the (B,H,W,Cout) map, or
    (slabs, n_slabs, slab_stride) when `want_slabs` (raw partial sums, no bias/epilogue).
    `groups`: `stats` is [groups][2][Cout], one statistics group per B/groups images.
    `frozen`: K is never split (the split count follows the batch size, and with it the order of the fp32 sums)."""
    _require_cuda(x0)
    B, H, W, C0 = x0.shape
    C1 = x1.shape[3] if x1 is not None else 0
    Cout, Cin, k, _ = weight.shape
    assert Cin == C0 + C1, (Cin, C0, C1)
    weight_mem(weight)
    M = B * H * W
    nchunks = k * k * _cdiv(Cin, 32)
    fused = act != ACT_NONE or scale is not None
    split = plan_split_k(M, Cout, nchunks) if (allow_split and not fused and not frozen) else 1
    if want_slabs:
        slabs = torch.empty((split, B, H, W, Cout), device=x0.device, dtype=torch.float32)
        if split == 1:
            conv_raw(FWD, x0, x1, weight, slabs, B=B, H=H, W=W, ksize=k, Cin=Cin, Cout=Cout, a_split=C0)
        else:
            conv_raw(FWD, x0, x1, weight, slabs, B=B, H=H, W=W, ksize=k, Cin=Cin, Cout=Cout, a_split=C0,
                     split_k=split, slab_stride=M * Cout)
        return slabs, split, M * Cout
    out = torch.empty((B, H, W, Cout), device=x0.device, dtype=torch.float32)
    if split == 1:
        conv_raw(FWD, x0, x1, weight, out, B=B, H=H, W=W, ksize=k, Cin=Cin, Cout=Cout, a_split=C0, act=act,
                 bias=bias, scale=scale, shift=shift, stats=stats,
                 stats_rows=(M // groups if (groups > 1 and stats is not None) else 0))
    else:
        slabs = torch.empty((split, M * Cout), device=x0.device, dtype=torch.float32)
        conv_raw(FWD, x0, x1, weight, slabs, B=B, H=H, W=W, ksize=k, Cin=Cin, Cout=Cout, a_split=C0,
                 split_k=split, slab_stride=M * Cout)
        call("rac_slab_reduce", ptr(slabs), split, M * Cout, ptr(bias), ptr(out), M * Cout, Cout, None, stream_ptr())
        if stats is not None:
            call("rac_col_stats", ptr(out), ptr(stats), M, Cout, groups, stream_ptr())
    return out


def conv_dgrad(dy, weight, C0: int, C1: int = 0, transposed_head: bool = False):
    """Data gradient of a conv whose input was the concat of C0 + C1 channels: returns (dx0, dx1|None)."""
    B, H, W, Cout = dy.shape
    Co, Cin, k, _ = weight.shape
    assert Co == Cout and Cin == C0 + C1
    M = B * H * W
    nchunks = k * k * _cdiv(Cout, 32)
    dx0 = torch.empty((B, H, W, C0), device=dy.device, dtype=torch.float32)
    dx1 = torch.empty((B, H, W, C1), device=dy.device, dtype=torch.float32) if C1 else None
    split = plan_split_k(M, Cin, nchunks)
    if split > 1:
        slabs = torch.empty((split, M * Cin), device=dy.device, dtype=torch.float32)
        conv_raw(DGRAD, dy, None, weight, slabs, B=B, H=H, W=W, ksize=k, Cin=Cin, Cout=Cout, split_k=split,
                 slab_stride=M * Cin)
        if C1:
            call("rac_slab_reduce2", ptr(slabs), split, M * Cin, None, ptr(dx0), ptr(dx1), M, Cin, C0, None, None,
                 stream_ptr())
        else:
            call("rac_slab_reduce", ptr(slabs), split, M * Cin, None, ptr(dx0), M * Cin, Cin, None, stream_ptr())
    else:
        conv_raw(DGRAD, dy, None, weight, dx0, dx1, B=B, H=H, W=W, ksize=k, Cin=Cin, Cout=Cout,
                 o_split=C0 if C1 else 0)
    return dx0, dx1


def conv_wgrad_acc(dy, x0, x1, weight):
    """weight.grad += dW  (x = virtual concat [x0 | x1]); split-K chosen by the library."""
    B, H, W, Cout = dy.shape
    Co, Cin, k, _ = weight.shape
    C0 = x0.shape[3]
    g = grad_buffer(weight)
    weight_mem(g)
    conv_raw(WGRAD, x0, x1, dy, g, B=B, H=H, W=W, ksize=k, Cin=Cin, Cout=Cout, a_split=C0, accumulate=1, split_k=0)


def bias_grad_acc(dy, bias):
    """bias.grad += column sums of dy.  Inside `deferred_wgrad()` the few-row tensors of the recurrent part are only
    recorded: one launch per bias sums all its time steps when the context exits."""
    M = dy.numel() // dy.shape[-1]
    if _DEFERRED is not None and M <= _COLSUM_STEPS_MAX_ROWS:
        _DEFERRED_BIAS.setdefault(id(bias), (bias, []))[1].append(dy)
        return
    call("rac_colsum_acc", ptr(dy), ptr(grad_buffer(bias)), ptr(_colsum_parts(dy.device, M, dy.shape[-1])), M,
         dy.shape[-1], stream_ptr())


# bias gradients are summed in two stages -- per row block, then the blocks in a fixed order -- so that a train step is
# bit-reproducible (one fp32 atomic per workgroup and column was the step's last order-dependent fp32 sum);
# RAC_COLSUM_ATOMIC=1: the atomics
COLSUM_ATOMIC = os.environ.get("RAC_COLSUM_ATOMIC", "0") == "1"


def _colsum_parts(device, M: int, Cc: int):
    if COLSUM_ATOMIC:
        return None
    return torch.empty(int(_lib.load().rac_colsum_blocks(M, Cc)) * Cc, device=device, dtype=torch.float32)


_COLSUM_STEPS_MAX_ROWS = 16384
_DEFERRED_BIAS = {}


def _flush_bias_grads():
    pending = dict(_DEFERRED_BIAS)
    _DEFERRED_BIAS.clear()
    sp = stream_ptr()
    for bias, dys in pending.values():
        g = grad_buffer(bias)
        Cc = dys[0].shape[-1]
        M = dys[0].numel() // Cc
        for lo in range(0, len(dys), _lib.WGRAD_MAX_STEPS):
            chunk = dys[lo:lo + _lib.WGRAD_MAX_STEPS]
            assert all(d.shape == dys[0].shape and d.is_contiguous() for d in chunk)
            xs = (C.c_void_p * len(chunk))(*[ptr(d) for d in chunk])
            call("rac_colsum_steps", xs, len(chunk), ptr(g), ptr(_colsum_parts(g.device, M, Cc)), M, Cc, sp)


# --------------------------------------------------------------------------- #
# split-precision convolutions on the fp16 matrix pipe (csrc/rac_split16.hip)
# --------------------------------------------------------------------------- #
# Every fp32 operand is scaled by a power of two taken from its max |x| (computed on the device: `amax` slots hold
# the bit pattern of the maximum) and split into two fp16 parts; three part-products per fp32 product, fp32
# accumulation.  RAC_SPLIT_GEMM=0 keeps every conv on the exact-fp32 MFMA path (rac_conv2d).
SPLIT_GEMM = os.environ.get("RAC_SPLIT_GEMM", "1") == "1"
SPLIT_GEMM_TRAIN = SPLIT_GEMM
HEAD_DIRECT = os.environ.get("RAC_HEAD_DIRECT", "1") == "1"  # the 64 -> 4 output head as FMAs (rac_head_fwd)
FIRST_MFMA = os.environ.get("RAC_FIRST_MFMA", "1") == "1"  # the frozen model's first encoder layer on the matrix pipe
HEAD_DGRAD = os.environ.get("RAC_HEAD_DGRAD", "1") == "1"  # its data gradient as one streaming pass (rac_head_dgrad)
HEAD_MFMA = os.environ.get("RAC_HEAD_MFMA", "1") == "1"  # ... on the matrix pipe, roles swapped (rac_head_fwd_split)
# narrowest layer (output channels) that runs split-precision
SPLIT_MIN_COUT = int(os.environ.get("RAC_SPLIT_MIN_COUT", "64"))
SPLIT_MIN_COUT_TRAIN = int(os.environ.get("RAC_SPLIT_MIN_COUT_TRAIN", "64"))

_AMAX = {"buf": None, "used": 0, "one": None}
_AMAX_SLOTS = 1 << 20


def _amax_take(device, n: int) -> torch.Tensor:
    """`n` zeroed slots (an int32 view) of the arena; a full arena is replaced by a fresh zeroed one (tensors that still
    carry slots of the old one keep it alive)."""
    buf = _AMAX["buf"]
    if buf is None or buf.device != torch.device(device) or _AMAX["used"] + n > _AMAX_SLOTS:
        buf = _AMAX["buf"] = torch.zeros(max(_AMAX_SLOTS, n), device=device, dtype=torch.int32)
        _AMAX["used"] = 0
    k = _AMAX["used"]
    _AMAX["used"] = k + n
    return buf[k:k + n]


def amax_of(x0: torch.Tensor, x1: Optional[torch.Tensor] = None, per_image: bool = False) -> torch.Tensor:
    """Device slot (1-element int32 view) holding the bit pattern of max |x| over x0 (and x1); `per_image`: one slot
    per leading index of x0 (B slots: the frozen model scales every image by its own maximum)."""
    if per_image:
        assert x1 is None
        slot = _amax_take(x0.device, x0.shape[0])
        call("rac_absmax_rows", ptr(x0), x0.shape[0], x0.numel() // x0.shape[0], ptr(slot), stream_ptr())
        return slot
    slot = _amax_take(x0.device, 1)
    call("rac_absmax", ptr(x0), x0.numel(), ptr(x1), x1.numel() if x1 is not None else 0, ptr(slot), stream_ptr())
    return slot


def amax_slot(device, n: int = 1) -> torch.Tensor:
    """Zeroed slot(s) for a kernel that folds the max |v| of its output in (`*_amax` output arguments of the C ABI);
    n = B: one per image."""
    return _amax_take(device, n)


def amax_one(device, n: int = 1) -> torch.Tensor:
    """Slot(s) for tensors bounded by 1 in magnitude (ConvLSTM hidden states h = o * tanh(c); frames and masks)."""
    one = _AMAX["one"]
    if one is None or one.device != torch.device(device) or one.numel() < n:
        one = _AMAX["one"] = torch.full((max(n, 1024),), 0x3F800000, device=device, dtype=torch.int32)
    view = one[:n]
    view._rac_bound1 = True  # amax_for hands out as many of these as the consumer's granularity needs
    return view


def tag_amax(t: torch.Tensor, slot: torch.Tensor) -> torch.Tensor:
    """Remember a tensor's max-|x| slot on the tensor object (producers that know a bound, or consumers that already
    measured it, save the next consumer a reduction pass)."""
    t._rac_amax = slot
    t._rac_amax_ver = t._version  # an in-place change afterwards (autograd accumulating a second gradient into this
    return t                      # buffer: `old.add_(new)`) bumps the version: the tag no longer describes the tensor


def amax_tag(t):
    if t is None or getattr(t, "_rac_amax_ver", None) != t._version:
        return None
    return getattr(t, "_rac_amax", None)


def retag(t, slot):
    """Saved tensors come back from autograd as new Python objects: put the slot measured in forward() back on."""
    if t is not None and slot is not None:
        t._rac_amax = slot
        t._rac_amax_ver = t._version
    return t


def amax_for(t: torch.Tensor, per_image: bool = False) -> torch.Tensor:
    """The tensor's slot (per_image: its B slots), measured now unless a producer left a tag of that kind."""
    slot = amax_tag(t)  # (None for a tensor modified since it was tagged)
    want = t.shape[0] if per_image else 1
    if slot is not None and getattr(slot, "_rac_bound1", False):
        return amax_one(t.device, want)  # |t| <= 1 whatever the granularity
    if slot is None or slot.numel() != want:
        slot = amax_of(t, per_image=per_image)
        tag_amax(t, slot)
    return slot


def broadcast_images(t: torch.Tensor, n: int) -> torch.Tensor:
    """A one-image map as n identical images, its per-image maximum carried along (the planner's first step: every
    candidate starts from the same frame, so the encoder runs once -- an image's result does not depend on its batch)."""
    out = t.expand(n, *t.shape[1:]).contiguous()
    slot = amax_tag(t)
    if slot is not None:
        if getattr(slot, "_rac_bound1", False):
            tag_amax(out, amax_one(t.device, n))
        elif slot.numel() == 1:
            tag_amax(out, slot.expand(n).contiguous())
    return out


_SPLIT_OK = {}


def split_supported(H: int, W: int, k: int, Cin: int, Cout: int, a_split: int = 0) -> bool:
    key = (H, W, k, Cin, Cout, a_split)
    ok = _SPLIT_OK.get(key)
    if ok is None:
        ok = _SPLIT_OK[key] = bool(_lib.load().rac_conv2d_split_supported(H, W, k, Cin, Cout, a_split))
    return ok


# Split-precision weight operands.  Every conv weight that runs on the fp16 pipe is registered here with its amax slot
# and its fragment-order parts (forward and / or transposed); after an optimiser step ALL of them are stale at once and
# are refreshed by three launches (zero the slots, rac_absmax_multi, rac_weight_frag_split_multi) instead of three per
# weight and direction.  A weight that goes stale on its own (a new registration) takes the single-tensor calls.
class _WeightParts:
    # slot: what the convs read as the weight's maximum (exact after a refresh here; an upper bound after a fused
    # optimiser step, fused_adam_step); exact_ok: the device's `exact` array holds the exact maximum of the CURRENT values
    __slots__ = ("ref", "idx", "slot", "device", "parts", "tag", "exact_ok", "__weakref__")


_WP_ENTRIES = {}  # id(weight) -> _WeightParts
_WP_DEV = {}      # device -> {"slots", "free", "sig", "tables"}
_WP_NSLOTS = 1024


def _wp_state(device):
    st = _WP_DEV.get(device)
    if st is None:
        st = _WP_DEV[device] = {"slots": torch.zeros(_WP_NSLOTS, device=device, dtype=torch.int32),
                                "exact": torch.zeros(_WP_NSLOTS, device=device, dtype=torch.int32),
                                "free": list(range(_WP_NSLOTS - 1, -1, -1)), "tables": {}, "adam_plan": None}
    return st


def _wp_drop(key, device, idx, _ref):
    ent = _WP_ENTRIES.get(key)
    if ent is not None and ent.idx == idx and ent.device == device:
        del _WP_ENTRIES[key]
        st = _WP_DEV.get(device)
        if st is not None:
            st["free"].append(idx)


def _wp_tag(weight):
    # a view over several parameters (the merged mu | logvar head) changes when any of them does: in-place updates of
    # a parameter bump ITS version counter, not the view's
    src = getattr(weight, "_rac_sources", None)
    ver = weight._version if src is None else tuple(t._version for t in src)
    return (ver, PARAM_EPOCH, weight.data_ptr())


def _wp_upload(jobs, device):
    raw = bytes(jobs)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


_ADAM_SIDE = {"stream": None}  # fused_adam_step(late=...): the late weights' update runs here
ADAM_LATE_WGS = int(os.environ.get("RAC_ADAM_LATE_WGS", "512"))  # ... on this many workgroups (2 per CU)
ADAM_LOW_PRIORITY = os.environ.get("RAC_ADAM_LOW_PRIORITY", "0") == "1"  # (experiments: a below-default-priority HIP stream)
ADAM_WGS = int(os.environ.get("RAC_ADAM_WGS", "0"))  # (experiments: the one-stream pass on a bounded grid too; 0 = one workgroup per block)


def low_priority_stream(dev):
    """A HIP stream BELOW the default priority (torch offers only default and higher): its workgroups take the slots the
    default-priority streams leave.  Falls back to a plain stream if the runtime refuses."""
    try:
        hip = C.CDLL("libamdhip64.so")
        lo, hi = C.c_int(0), C.c_int(0)
        if hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)) == 0 and lo.value > 0:
            with torch.cuda.device(dev):
                h = C.c_void_p()
                if hip.hipStreamCreateWithPriority(C.byref(h), C.c_uint(1), C.c_int(lo.value)) == 0 and h.value:  # 1: non-blocking
                    return torch.cuda.ExternalStream(h.value, device=dev)
    except OSError:
        pass
    return torch.cuda.Stream(device=dev)

# optim.ShardedAdam while the all-gather of the updated parameters is in flight (or optim.FusedAdam while its late weights'
# update runs on the side stream): `ready(param)` says whether a parameter's
# buckets have been waited for (its new values may be read by kernels enqueued now).  _wp_refresh leaves the others stale.
PARAM_GATE = None


def param_wait(t: torch.Tensor = None) -> None:
    """Make the current stream wait for an optimiser update that is still in flight (PARAM_GATE): of the parameter memory
    `t`, or of everything."""
    gate = PARAM_GATE
    if gate is None:
        return
    if t is None:
        gate.wait_params()
    elif hasattr(gate, "wait_for"):
        gate.wait_for(t)
    elif not gate.ready(t):
        gate.wait_params()


def _wp_refresh(device):
    """Bring every registered weight of `device` whose parameter changed up to date."""
    st = _wp_state(device)
    stale, n_live = [], 0
    gate = PARAM_GATE
    for ent in list(_WP_ENTRIES.values()):
        if ent.device != device:
            continue
        w = ent.ref()
        if w is None:
            continue
        src = getattr(w, "_rac_pad_source", None)
        if gate is not None:  # (a padded copy, a merged head view: judged by the parameter memory they alias)
            base = src[0]() if src is not None else w
            if base is not None and not gate.ready(base):
                continue
        if src is not None:  # a zero-padded copy of a parameter: rebuild the copy first
            param = src[0]()
            if param is None:
                continue
            padded_weight(param, src[1])
        n_live += 1
        if ent.tag != _wp_tag(w):
            stale.append((ent, w))
    if not stale:
        return
    sp = stream_ptr()
    lib = _lib.load()
    if len(stale) <= 2:  # a new registration: the single-tensor calls
        for ent, w in stale:
            ent.slot.zero_()
            wm = weight_mem(w.detach())
            co, ci, k, _ = w.shape
            call("rac_absmax", ptr(wm), wm.numel(), None, 0, ptr(ent.slot), sp)
            for transposed, parts in ent.parts.items():
                call("rac_weight_frag_split", ptr(wm), ptr(ent.slot), ptr(parts), co, ci, k, 1 if transposed else 0,
                     wm.numel(), sp)
            st["exact"][ent.idx:ent.idx + 1].copy_(ent.slot)
            ent.tag, ent.exact_ok = _wp_tag(w), True
        return
    sig = tuple((w.data_ptr(), ent.idx, tuple(sorted((t, q.data_ptr()) for t, q in ent.parts.items())))
                for ent, w in stale)
    tables = st["tables"].get(sig)
    if tables is None:  # job tables (device memory) for this set of weights
        if len(st["tables"]) > 8:
            st["tables"].clear()
        ajobs = (AbsmaxJob * len(stale))()
        nfrag = sum(len(ent.parts) for ent, _ in stale)
        fjobs = (FragJob * nfrag)()
        ab = fb = j = 0
        for i, (ent, w) in enumerate(stale):
            wm = weight_mem(w.detach())
            co, ci, k, _ = w.shape
            ajobs[i] = AbsmaxJob(x=ptr(wm), n=wm.numel(), amax=ptr(ent.slot), block_begin=ab)
            ab += lib.rac_absmax_blocks(wm.numel())
            for transposed, parts in sorted(ent.parts.items()):
                fjobs[j] = FragJob(w=ptr(wm), w_amax=ptr(ent.slot), parts=ptr(parts), part_stride=wm.numel(), Cout=co,
                                   Cin=ci, ksize=k, transposed=1 if transposed else 0, block_begin=fb)
                fb += lib.rac_weight_frag_blocks(co, ci, k)
                j += 1
        idx = torch.tensor([ent.idx for ent, _ in stale], device=device, dtype=torch.long)
        tables = st["tables"][sig] = (_wp_upload(ajobs, device), len(stale), ab, _wp_upload(fjobs, device), nfrag, fb, idx)
    ta, na, ab, tf, nf, fb, idx = tables
    st["slots"].index_fill_(0, idx, 0)  # the stale slots are recomputed (the others may hold bounds of a fused step)
    call("rac_absmax_multi", ptr(ta), na, ab, sp)
    call("rac_weight_frag_split_multi", ptr(tf), nf, fb, sp)
    st["exact"].index_copy_(0, idx, st["slots"].index_select(0, idx))
    for ent, w in stale:
        ent.tag, ent.exact_ok = _wp_tag(w), True


# Fused optimiser step: for the conv weights that live in the model's flat parameter buffer and run on the split-precision
# pipe, ONE pass does Adam and writes the next step's fp16 fragment parts (rac_adam_frag_multi) -- instead of Adam, then
# a pass for the maxima, then one that re-reads every weight to split it (three reads of the 954 MB at g 512).  The
# parts must be scaled before the new maximum is known: Adam moves an element by at most lr * adam_step_bound(), so
# max |w| + that margin bounds the new maximum, and its exponent is the scale; the exact maximum comes out of the same
# pass for the next step's bound.  Everything else in the flat buffer takes plain Adam over the complementary ranges.
ADAM_FUSED = os.environ.get("RAC_ADAM_FUSED", "1") == "1"
_ADAM_BOUNDS = {}


def adam_step_bound(beta1: float, beta2: float):
    """sup over t of |m_hat_t| / sqrt(v_hat_t) for ANY gradient history (Cauchy-Schwarz over the two exponential
    averages): the largest multiple of lr an Adam step can move an element.  None when it is unbounded."""
    key = (beta1, beta2)
    if key not in _ADAM_BOUNDS:
        r = beta1 * beta1 / beta2
        if not (0.0 <= beta1 < 1.0 and 0.0 < beta2 < 1.0 and r < 1.0):
            _ADAM_BOUNDS[key] = None
        else:
            best, series, b1t, b2t, rk = 0.0, 0.0, 1.0, 1.0, 1.0
            for _ in range(200000):
                series += rk
                rk *= r
                b1t *= beta1
                b2t *= beta2
                best = max(best, (1 - beta1) ** 2 / (1 - b1t) ** 2 * (1 - b2t) / (1 - beta2) * series)
                if rk < 1e-18 and b1t < 1e-18 and b2t < 1e-9:
                    break
            _ADAM_BOUNDS[key] = best ** 0.5
    return _ADAM_BOUNDS[key]


def fused_adam_step(flat, grad, m, v, lr, beta1, beta2, eps, step, late=None):
    """One optimiser step over the flat buffers with the registered split-precision weights' parts refreshed in the
    same pass.  False (nothing launched) when that is not possible yet -- a weight whose parts or maximum are not
    current (first steps, new registrations), no registered weight inside `flat`, an unbounded Adam step: the caller
    then takes rac_adam_step and the parts are refreshed lazily.

    `late` = (first flat element, {data_ptr of the weights that may be late} or a list of such sets, in the order the next
    forward pass needs them): the registered weights behind that element whose memory is in a set are updated on a SIDE
    stream (behind everything enqueued so far; one launch per set, in order) and the call returns
    [(event, [(offset, numel)]) per set] -- the event the consumer of those weights must wait for -- instead of True: the pass is
    bound by HBM (8.6 GB per step), the next step's encoder forward is many small launches bound by latency; the caller
    (optim.FusedAdam with `overlap_next_forward`) makes the model wait behind its encoder."""
    global PARAM_EPOCH
    if not ADAM_FUSED:
        return False
    bound = adam_step_bound(beta1, beta2)
    if bound is None:
        return False
    dev = flat.device
    st = _wp_state(dev)
    base, nbytes = flat.data_ptr(), flat.numel() * 4
    covered = []
    for ent in _WP_ENTRIES.values():
        if ent.device != dev:
            continue
        w = ent.ref()
        if w is None or getattr(w, "_rac_pad_source", None) is not None:
            continue
        off = w.data_ptr() - base
        co, ci, k, _ = w.shape
        if not (0 <= off < nbytes and off % 16 == 0 and w.stride() == (k * k * ci, 1, k * ci, ci)):
            continue
        if ent.tag != _wp_tag(w) or not ent.exact_ok:
            return False
        covered.append((off // 4, ent, w))
    if not covered:
        return False
    covered.sort(key=lambda c: c[0])
    # late[1]: one set of data pointers, or a LIST of sets = groups in the order the next forward pass needs them (each
    # group its own launch and event on the side stream, which runs them in that order)
    late_sets = [] if late is None else (list(late[1]) if isinstance(late[1], (list, tuple)) else [late[1]])

    def late_group(off, w):  # index of the weight's late group, -1: updated on the caller's stream
        if late is None or off < late[0]:
            return -1
        for gi, ptrs in enumerate(late_sets):
            if w.data_ptr() in ptrs:
                return gi
        return -1
    is_late = lambda off, w: late_group(off, w) >= 0
    sig = (base, grad.data_ptr(), m.data_ptr(), v.data_ptr(),
           None if late is None else (late[0], tuple(tuple(sorted(ps)) for ps in late_sets)),
           tuple((off, ent.idx, tuple(sorted((t, q.data_ptr()) for t, q in ent.parts.items()))) for off, ent, _ in covered))
    plan = st["adam_plan"]
    if plan is not None and plan["sig"] == sig and plan.get("unsupported"):
        return False
    if plan is None or plan["sig"] != sig:
        lib = _lib.load()
        jobs = (AdamFragJob * len(covered))()
        groups = [late_group(off, w) for off, _, w in covered]
        n_early = sum(1 for gi in groups if gi < 0)
        n_in = [sum(1 for gi in groups if gi == k) for k in range(len(late_sets))]
        first = [n_early + sum(n_in[:k]) for k in range(len(late_sets))]  # a group's first slot in the job table
        slot_of, taken = {}, [0] * len(late_sets)  # the early jobs first, then group by group (each numbers its own blocks)
        ne = 0
        for j, gi in enumerate(groups):
            if gi >= 0:
                slot_of[j], taken[gi] = first[gi] + taken[gi], taken[gi] + 1
            else:
                slot_of[j], ne = ne, ne + 1
        ranges, blocks, pos = [], 0, 0
        late_blocks = [0] * len(late_sets)
        late_spans = [[] for _ in late_sets]
        for i, (off, ent, w) in enumerate(covered):
            co, ci, k, _ = w.shape
            n = w.numel()
            if off % 4 or n % 4 or off < pos or flat.numel() % 4:
                # registered views that overlap (the merged mu | logvar head next to its own two parameters) or are not
                # 16-byte aligned: this layout is not the fused pass's -- plain Adam, parts refreshed lazily
                st["adam_plan"] = {"sig": sig, "unsupported": True}
                return False
            if off > pos:
                ranges.append((pos // 4, (off - pos) // 4))
            pos = off + n
            gi = groups[i]
            lt = gi >= 0
            jobs[slot_of[i]] = AdamFragJob(p=base + 4 * off, g=grad.data_ptr() + 4 * off, m=m.data_ptr() + 4 * off,
                                           v=v.data_ptr() + 4 * off, scale_slot=ptr(ent.slot),
                                           amax_out=ptr(st["exact"][ent.idx:ent.idx + 1]), parts_fwd=ptr(ent.parts.get(False)),
                                           parts_t=ptr(ent.parts.get(True)), part_stride=n, Cout=co, Cin=ci, ksize=k, reserved=0,
                                           block_begin=late_blocks[gi] if lt else blocks)
            if lt:
                late_blocks[gi] += lib.rac_weight_frag_blocks(co, ci, k)
                late_spans[gi].append((off, n))
            else:
                blocks += lib.rac_weight_frag_blocks(co, ci, k)
        if pos < flat.numel():
            ranges.append((pos // 4, (flat.numel() - pos + 3) // 4))
        rjobs = (AdamRange * max(1, len(ranges)))()
        rblocks = 0
        for i, (b4, n4) in enumerate(ranges):
            rjobs[i] = AdamRange(begin4=b4, n4=n4, block_begin=rblocks)
            rblocks += _cdiv(n4, 1024)
        plan = st["adam_plan"] = {
            "sig": sig, "jobs": _wp_upload(jobs, dev), "n_jobs": len(covered), "blocks": blocks,
            "n_early": n_early, "late_blocks": late_blocks, "late_spans": late_spans, "late_first": first, "late_n": n_in,
            "ranges": _wp_upload(rjobs, dev), "n_ranges": len(ranges), "rblocks": rblocks,
            "idx": torch.tensor([ent.idx for _, ent, _ in covered], device=dev, dtype=torch.int32)}
    sp = stream_ptr()
    call("rac_amax_bound", ptr(st["exact"]), ptr(st["slots"]), ptr(plan["idx"]), plan["n_jobs"],
         float(lr) * bound * 1.01, sp)
    n_early, n_late = plan["n_early"], plan["n_jobs"] - plan["n_early"]
    if n_early and ADAM_WGS:
        call("rac_adam_frag_multi_bounded", ptr(plan["jobs"]), n_early, plan["blocks"], ADAM_WGS, float(lr), float(beta1),
             float(beta2), float(eps), int(step), sp)
    elif n_early:
        call("rac_adam_frag_multi", ptr(plan["jobs"]), n_early, plan["blocks"], float(lr), float(beta1), float(beta2),
             float(eps), int(step), sp)
    if plan["n_ranges"]:
        call("rac_adam_ranges", ptr(flat), ptr(grad), ptr(m), ptr(v), ptr(plan["ranges"]), plan["n_ranges"],
             plan["rblocks"], float(lr), float(beta1), float(beta2), float(eps), int(step), sp)
    done = []
    if n_late:
        main = torch.cuda.current_stream()
        if _ADAM_SIDE["stream"] is None or _ADAM_SIDE["stream"].device != dev:
            _ADAM_SIDE["stream"] = low_priority_stream(dev) if ADAM_LOW_PRIORITY else torch.cuda.Stream(device=dev)
        side = _ADAM_SIDE["stream"]
        ready = torch.cuda.Event()
        ready.record(main)  # every gradient, the scale bounds and whatever read the old weights precede the side launch
        side.wait_event(ready)
        with torch.cuda.stream(side):
            for gi, n_g in enumerate(plan["late_n"]):
                if not n_g:
                    continue
                call("rac_adam_frag_multi_bounded", ptr(plan["jobs"]) + plan["late_first"][gi] * C.sizeof(AdamFragJob), n_g,
                     plan["late_blocks"][gi], ADAM_LATE_WGS, float(lr), float(beta1), float(beta2), float(eps), int(step),
                     stream_ptr())
                ev = torch.cuda.Event()
                ev.record(side)
                done.append((ev, list(plan["late_spans"][gi])))
    PARAM_EPOCH += 1
    for _, ent, w in covered:  # their parts and maxima already describe the new values
        ent.tag = _wp_tag(w)
    return done if done else True


def weight_parts(weight: torch.Tensor, transposed: bool = False):
    """(fp16 parts of weight * 2^k in MFMA fragment order, amax slot), kept until the parameter changes;
    `transposed`: of the (Cin, Cout) tap-flipped weight whose forward conv is the data gradient."""
    key = id(weight)
    ent = _WP_ENTRIES.get(key)
    if ent is None or ent.ref() is not weight or ent.device != weight.device:
        st = _wp_state(weight.device)
        if not st["free"]:
            raise _lib.RacError("too many conv weights registered for the split-precision pipe")
        co, ci, k, _ = weight.shape
        if co % 32 or ci % 32:
            raise _lib.RacError(f"split-precision weight {tuple(weight.shape)}: channel counts must be multiples of 32")
        ent = _WeightParts()
        ent.idx = st["free"].pop()
        ent.device, ent.parts, ent.tag, ent.exact_ok = weight.device, {}, None, False
        ent.slot = st["slots"][ent.idx:ent.idx + 1]
        ent.ref = weakref.ref(weight, functools.partial(_wp_drop, key, weight.device, ent.idx))
        _WP_ENTRIES[key] = ent
    if transposed not in ent.parts:
        parts = torch.empty((2, weight.numel()), device=weight.device, dtype=torch.float16)
        parts._rac_transposed = transposed
        ent.parts[transposed] = parts
        ent.tag = None
    if ent.tag != _wp_tag(weight):
        gate = PARAM_GATE
        if gate is not None:  # this weight's new values may still be travelling (a sharded optimiser's all-gather)
            src = getattr(weight, "_rac_pad_source", None)
            base = src[0]() if src is not None else weight
            if base is not None and not gate.ready(base):
                gate.wait_params()
        _wp_refresh(weight.device)
    return ent.parts[transposed], ent.slot


_CONV_ARGS = {}  # shape key -> the ConvArgs struct re-pointed per launch (_split_launch)


def _split_launch(x0, x1, a0, a1, pw, wslot, out, *, B, H, W, k, Cin, Cout, C0, act=ACT_NONE, bias=None, scale=None,
                  shift=None, stats=None, split_k=1, slab_stride=0, stats_rows=0, out_amax=None, w_cin=0, a0_up=0,
                  per_image=False, pool=None):
    """`pool`: None, a [B, H/2, W/2, Cout] tensor the same launch fills with the 2 x 2 max pool of its output, or "query"
    (nothing launched: can this conv pool in its epilogue?  -> bool)."""
    # (the argument struct of a shape is built once and re-pointed: the library reads it during the call only; building a
    # 25-field ctypes structure from keywords costs more host time than the launch itself)
    key = (B, H, W, k, Cin, Cout, act, split_k, C0, slab_stride, stats_rows, a0_up, per_image)
    args = _CONV_ARGS.get(key)
    if args is None:
        args = _CONV_ARGS[key] = ConvArgs(mode=FWD, B=B, H=H, W=W, ksize=k, Cin=Cin, Cout=Cout, act=act, split_k=split_k,
                                          accumulate=0, a_split=C0, o_split=0, slab_stride=slab_stride, stats_rows=stats_rows,
                                          a0_up=a0_up, amax_per_image=1 if per_image else 0)
    args.a0, args.a1, args.w, args.out0 = ptr(x0), ptr(x1), ptr(pw), ptr(out)
    args.out1 = None if isinstance(pool, str) else ptr(pool)
    args.bias, args.scale, args.shift, args.stats = ptr(bias), ptr(scale), ptr(shift), ptr(stats)
    if isinstance(pool, str):
        return bool(_lib.load().rac_conv2d_fwd_split_pool_ok(C.byref(args), w_cin))
    prof = PROFILE
    timed = prof is not None and prof["match"] == (FWD, k, Cin, Cout)
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if SHAPE_LOG is not None:
        _log_shape("conv16", DGRAD if getattr(pw, "_rac_transposed", False) else FWD, k, B * H * W, Cout, Cin * k * k,
                   2500.0 / 3)
    call("rac_conv2d_fwd_split", C.byref(args), ptr(a0), ptr(a1), pw.shape[1], w_cin, ptr(wslot), ptr(out_amax),
         stream_ptr())
    if timed:
        e1.record()
        prof["events"].append((e0, e1, B * H * W))
        prof["split"] = True


def is_zero(t) -> bool:
    """Tensors known to be all zeros (a ConvLSTM's initial state): convs skip their share of K."""
    return t is not None and getattr(t, "_rac_zero", False)


def per_image_ok(H: int, W: int) -> bool:
    """Map sizes whose per-image scales the split kernels take (whole 16-row blocks per image)."""
    return H * W > 128 or (H * W) % 16 == 0


SLAB_STATS = os.environ.get("RAC_SLAB_STATS", "1") == "1"  # split-K combine + BatchNorm statistics in one pass


POOL_FUSED = os.environ.get("RAC_POOL_FUSED", "1") == "1"  # frozen encoder: MaxPool2d in the producing conv's epilogue
_POOL_OK = {}  # (B, H, W, C0, C1, Cout, k, per_image) -> can this conv pool in its epilogue?


def conv_forward_split(x0, x1, weight, bias=None, *, act=ACT_NONE, scale=None, shift=None, stats=None,
                       want_slabs=False, groups=1, x0_up=False, per_image=False, pool=False):
    """FWD conv over [x0 | x1] on the fp16 matrix pipe with fp32-level accuracy (see include/rac_hip.h).
    `want_slabs`: raw split-K partial sums (slabs, n_slabs, slab_stride) for the ConvLSTM cell kernel.
    An all-zero x1 (`is_zero`) is skipped: the conv runs over the x0 channels of the same weight parts.
    `per_image` (the frozen model): every image is scaled by its own max |x| and K is never split, so an image's
    result is the same bits whatever else is in the batch and however large the batch is."""
    _require_cuda(x0)
    B, H, W, C0 = x0.shape
    if x0_up:  # x0 is the half-resolution tensor; its nearest 2x upsampling is the conv's first source
        H, W = 2 * H, 2 * W
        assert x1 is None or tuple(x1.shape[:3]) == (B, H, W)
    Cout, Cin_w, k, _ = weight.shape
    w_cin = 0
    if is_zero(x1) and C0 % 32 == 0:
        x1, w_cin = None, Cin_w
    C1 = x1.shape[3] if x1 is not None else 0
    Cin = C0 + C1
    assert Cin == Cin_w or w_cin
    M = B * H * W
    x0 = x0 if x0.is_contiguous() else x0.contiguous()
    if x1 is not None and not x1.is_contiguous():
        x1 = x1.contiguous()
    a0 = amax_for(x0, per_image)
    a1 = amax_for(x1, per_image) if x1 is not None else None
    pw, wslot = weight_parts(weight)
    kw = dict(B=B, H=H, W=W, k=k, Cin=Cin, Cout=Cout, C0=C0, w_cin=w_cin, a0_up=1 if x0_up else 0,
              per_image=per_image)
    if want_slabs:
        split = 1 if per_image else plan_split_k(M, Cout, k * k * _cdiv(Cin, 32), tile128_only=True)
        out = torch.empty((split, B, H, W, Cout), device=x0.device, dtype=torch.float32)
        _split_launch(x0, x1, a0, a1, pw, wslot, out, split_k=split, slab_stride=M * Cout, **kw)
        return out, split, M * Cout
    out = torch.empty((B, H, W, Cout), device=x0.device, dtype=torch.float32)
    fused = act != ACT_NONE or scale is not None
    split = 1 if (fused or per_image) else plan_split_k(M, Cout, k * k * _cdiv(Cin, 32), tile128_only=True)
    # the kernel that writes `out` also leaves its max |v| (per image: maxima) for the next conv
    slot = amax_slot(x0.device, B if per_image else 1)
    if pool:
        # (out, MaxPool2d(2)(out)) -- vgg_64.py:104-129 -- the pool written by the conv's own epilogue where the kernel can
        # (one read of `out` and one launch less), by rac_maxpool2_fwd otherwise: the same bits either way
        assert split == 1 and stats is None and not x0_up
        key = (B, H, W, C0, C1, Cout, k, per_image, w_cin)
        ok = _POOL_OK.get(key)
        if ok is None:
            ok = _POOL_OK[key] = POOL_FUSED and _split_launch(x0, x1, a0, a1, pw, wslot, out, act=act, bias=bias, scale=scale,
                                                               shift=shift, out_amax=slot, pool="query", **kw)
        pooled = torch.empty((B, H // 2, W // 2, Cout), device=x0.device, dtype=torch.float32)
        try:
            _split_launch(x0, x1, a0, a1, pw, wslot, out, act=act, bias=bias, scale=scale, shift=shift, out_amax=slot,
                          pool=pooled if ok else None, **kw)
        except _lib.RacError as e:
            # the library's tile plan changed under the cached answer (an A/B switch flipped in this process): separate pass
            if not ok or "cannot pool" not in str(e):
                raise
            ok = _POOL_OK[key] = False
            _split_launch(x0, x1, a0, a1, pw, wslot, out, act=act, bias=bias, scale=scale, shift=shift, out_amax=slot, **kw)
        if not ok:
            call("rac_maxpool2_fwd", ptr(out), ptr(pooled), B, H, W, Cout, stream_ptr())
        return tag_amax(out, slot), tag_amax(pooled, slot)  # max |pooled| <= max |out|: the same slot(s) bound both
    if split == 1:
        _split_launch(x0, x1, a0, a1, pw, wslot, out, act=act, bias=bias, scale=scale, shift=shift, stats=stats,
                      stats_rows=(M // groups if (groups > 1 and stats is not None) else 0), out_amax=slot, **kw)
    else:
        slabs = torch.empty((split, M * Cout), device=x0.device, dtype=torch.float32)
        _split_launch(x0, x1, a0, a1, pw, wslot, slabs, split_k=split, slab_stride=M * Cout, **kw)
        if (SLAB_STATS and stats is not None and bias is None and Cout % 4 == 0 and (Cout // 4) & (Cout // 4 - 1) == 0
                and Cout <= 1024):
            # a train-mode vgg layer: the combine pass also leaves the BatchNorm batch statistics (one read of the slabs)
            call("rac_slab_reduce_stats", ptr(slabs), split, M * Cout, ptr(out), ptr(stats), M, Cout, groups, ptr(slot),
                 stream_ptr())
            return tag_amax(out, slot)
        call("rac_slab_reduce", ptr(slabs), split, M * Cout, ptr(bias), ptr(out), M * Cout, Cout, ptr(slot),
             stream_ptr())
        if stats is not None:
            call("rac_col_stats", ptr(out), ptr(stats), M, Cout, groups, stream_ptr())
    return tag_amax(out, slot)


# The frozen model's ConvLSTM cell in ONE launch: gate conv + cell arithmetic in its epilogue (rac_convlstm_cell_fwd_split)
FUSED_CELL = os.environ.get("RAC_FUSED_CELL", "1") == "1"


def fused_cell_ok(x, weight) -> bool:
    B, H, W, g = x.shape
    return (FUSED_CELL and SPLIT_GEMM and H * W <= 128 and (H * W) % 16 == 0 and g % 16 == 0
            and tuple(weight.shape[:2]) == (4 * g, 2 * g) and split_supported(H, W, weight.shape[2], 2 * g, 4 * g, g))


def convlstm_cell_frozen(x, h_prev, c_prev, weight, bias):
    """(h, c) of a ConvLSTM cell without a tape (lstm.py:129-149): the gate conv over [x | h_prev] with per-image operand
    scales and the cell arithmetic in its epilogue, against a cached copy of the gate weight whose rows are
    gate-interleaved in groups of 16 channels (a lane of the matrix-pipe tile then holds i, f, o, g of one channel).  The
    [B, H, W, 4g] gate tensor is never written."""
    B, H, W, g = x.shape
    k = weight.shape[2]

    def build():
        w = weight.detach()
        wp = w.reshape(4, g // 16, 16, 2 * g, k, k).permute(1, 0, 2, 3, 4, 5).reshape(4 * g, 2 * g, k, k)
        return wp.contiguous(memory_format=torch.channels_last)
    w_perm = _derived(weight, "_rac_gate_interleaved", build)
    x = x if x.is_contiguous() else x.contiguous()
    x1, w_cin = h_prev, 0
    if is_zero(h_prev):  # first step: the hidden half of K is skipped (a channel prefix of the same parts)
        x1, w_cin = None, 2 * g
    elif not x1.is_contiguous():
        x1 = x1.contiguous()
    Cin = g if x1 is None else 2 * g
    a0 = amax_for(x, True)
    a1 = amax_for(x1, True) if x1 is not None else None
    pw, wslot = weight_parts(w_perm)
    c_prev = c_prev if c_prev.is_contiguous() else c_prev.contiguous()
    h, c = torch.empty_like(x), torch.empty_like(x)
    args = ConvArgs(mode=FWD, B=B, H=H, W=W, ksize=k, Cin=Cin, Cout=4 * g, act=ACT_NONE, split_k=1, accumulate=0, a_split=g,
                    o_split=0, slab_stride=0, a0=ptr(x), a1=ptr(x1), w=ptr(pw), out0=None, out1=None, bias=ptr(bias),
                    scale=None, shift=None, stats=None, stats_rows=0, a0_up=0, amax_per_image=1)
    prof = PROFILE
    timed = prof is not None and prof["match"] == (FWD, k, Cin, 4 * g)
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if SHAPE_LOG is not None:
        _log_shape("conv16", FWD, k, B * H * W, 4 * g, Cin * k * k, 2500.0 / 3)
    call("rac_convlstm_cell_fwd_split", C.byref(args), ptr(a0), ptr(a1), pw.shape[1], w_cin, ptr(wslot), ptr(c_prev),
         ptr(h), ptr(c), stream_ptr())
    if timed:
        e1.record()
        prof["events"].append((e0, e1, B * H * W))
        prof["split"] = True
    return h, c


def conv_dgrad_split(dy, weight, C0: int, C1: int = 0, need1: bool = True):
    """Data gradient on the split-precision pipe: forward conv of dy with the transposed, tap-flipped weight.
    `need1` False: only the first C0 input channels' gradient is computed (returns (dx0, None))."""
    B, H, W, Cout = dy.shape
    Co, Cin, k, _ = weight.shape
    assert Co == Cout and Cin == C0 + C1
    if C1 and not need1 and C0 % 32 == 0:
        Cin, C1 = C0, 0  # the first C0 rows of the transposed weight are a prefix of its parts
    M = B * H * W
    dy = dy if dy.is_contiguous() else dy.contiguous()
    pw, wslot = weight_parts(weight, transposed=True)
    split = plan_split_k(M, Cin, k * k * _cdiv(Cout, 32), tile128_only=True)
    dx0 = torch.empty((B, H, W, C0), device=dy.device, dtype=torch.float32)
    s0 = amax_slot(dy.device)  # the kernel that writes dx leaves max |dx| for the conv that consumes the gradient
    if split == 1 and not C1:  # one K range, one destination: the conv writes dx itself (no slab, no combine pass)
        _split_launch(dy, None, amax_for(dy), None, pw, wslot, dx0, B=B, H=H, W=W, k=k, Cin=Cout, Cout=Cin, C0=Cout,
                      out_amax=s0)
        return tag_amax(dx0, s0), None
    slabs = torch.empty((split, M * Cin), device=dy.device, dtype=torch.float32)
    _split_launch(dy, None, amax_for(dy), None, pw, wslot, slabs, B=B, H=H, W=W, k=k, Cin=Cout, Cout=Cin, C0=Cout,
                  split_k=split, slab_stride=M * Cin)
    if C1:
        dx1 = torch.empty((B, H, W, C1), device=dy.device, dtype=torch.float32)
        s1 = amax_slot(dy.device)
        call("rac_slab_reduce2", ptr(slabs), split, M * Cin, None, ptr(dx0), ptr(dx1), M, Cin, C0, ptr(s0), ptr(s1),
             stream_ptr())
        return tag_amax(dx0, s0), tag_amax(dx1, s1)
    call("rac_slab_reduce", ptr(slabs), split, M * Cin, None, ptr(dx0), M * Cin, Cin, ptr(s0), stream_ptr())
    return tag_amax(dx0, s0), None


# Deferred, time-batched weight gradients: inside `deferred_wgrad()` the split-precision wgrad of a weight that is
# applied at every time step (the ConvLSTM gate convs, the input convs) is not launched per step; its (dy, x0, x1)
# triples are kept and ONE launch over all steps' pixels runs when the context exits (one read-modify-write of the
# gradient instead of T, longer K loops).  Same sum, different association across steps.
_DEFERRED = None
DEFER_WGRAD = os.environ.get("RAC_DEFER_WGRAD", "1") == "1"
WGRAD_PRESPLIT = os.environ.get("RAC_WGRAD_PRESPLIT", "1") == "1"
# workgroups that stage one operand tile before splitting it once pays
WGRAD_PRESPLIT_MIN_READERS = int(os.environ.get("RAC_WGRAD_PRESPLIT_MIN", "32"))
# thin 3x3 layers on 32x32 / 64x64 maps: one workgroup keeps all nine taps (dy and x fetched once per tile)
WGRAD_ALLKY = os.environ.get("RAC_WGRAD_ALLKY", "1") == "1"


# The recurrent core's time-batched weight gradients (5.4 ms of matrix-pipe work at cfg2: 90 % of the parameters) do not
# wait for the end of the backward pass: their operands are complete when the core's backward is, so they are launched
# there on a SIDE STREAM and run under the rest of the backward pass (the encoder's: BatchNorm passes, pooling, small
# convs -- memory-bound kernels whose workgroups fit beside the weight-gradient kernels' two per CU).  The main stream
# waits for them before anything reads a gradient (all-reduce, Adam).  RAC_WGRAD_STREAM=0: on the main stream, at the end.
WGRAD_STREAM = os.environ.get("RAC_WGRAD_STREAM", "1") == "1"
WGRAD_CHAIN_FLUSH = os.environ.get("RAC_WGRAD_CHAIN_FLUSH", "1") == "1"  # per ConvLSTM chain (0: once, behind the core)
_SIDE = {"stream": None, "done": None, "keep": [], "on_ready": None}
WGRAD_LOW_PRIORITY = os.environ.get("RAC_WGRAD_LOW_PRIORITY", "0") == "1"  # (experiment: the side stream below default priority)
WGRAD_CU_MASK = os.environ.get("RAC_WGRAD_CU_MASK", "")  # (experiment: hex CU mask of the side stream, e.g. 3/4 of every XCD)


def _side_stream(dev):
    """The stream the weight gradients run on, beside the data-gradient chain."""
    st = _SIDE["stream"]
    if st is None or st.device != torch.device(dev):
        st = None
        if WGRAD_CU_MASK:
            try:
                hip = C.CDLL("libamdhip64.so")
                words = [int(WGRAD_CU_MASK[max(0, i - 8):i], 16) for i in range(len(WGRAD_CU_MASK), 0, -8)]
                arr = (C.c_uint32 * len(words))(*words)
                with torch.cuda.device(dev):
                    h = C.c_void_p()
                    if hip.hipExtStreamCreateWithCUMask(C.byref(h), C.c_uint32(len(words)), arr) == 0 and h.value:
                        st = torch.cuda.ExternalStream(h.value, device=dev)
            except (OSError, AttributeError):
                st = None
        if st is None:
            st = low_priority_stream(dev) if WGRAD_LOW_PRIORITY else torch.cuda.Stream(device=dev)
        _SIDE["stream"] = st
    return st


def _amax_reserve(device, n: int) -> None:
    """Make sure `n` more amax slots fit the current arena (a fresh arena is zero-filled by a kernel on the CURRENT stream:
    it must not be created while a side stream is current and then be used by the main stream's kernels)."""
    if _AMAX["buf"] is None or _AMAX["buf"].device != torch.device(device) or _AMAX["used"] + n > _AMAX_SLOTS:
        _amax_take(device, 0)
        if _AMAX["used"] + n > _AMAX_SLOTS:
            _AMAX["buf"] = torch.zeros(_AMAX_SLOTS, device=device, dtype=torch.int32)
            _AMAX["used"] = 0


def flush_deferred_wgrads_early(weights=None, final=True) -> None:
    """Launch the recorded (deferred) weight gradients of `weights` (None: all recorded so far), and the bias column sums
    recorded so far, NOW on the side stream.  Called where their last operand was produced (RecurrentCore.backward).
    `final`: nothing more will be recorded for these weights -- `on_ready` (the data-parallel reduction of a weight's
    gradient slice) may start behind the launch; a weight that records again afterwards raises."""
    if _DEFERRED is None or not WGRAD_STREAM or (weights is not None and not WGRAD_CHAIN_FLUSH):
        return
    if weights is None:  # everything recorded so far (each record's operands exist: they were produced in program order)
        items = list(_DEFERRED.values())
        bias_only = not items and bool(_DEFERRED_BIAS)
        _DEFERRED.clear()
    else:
        items = [_DEFERRED.pop(id(w)) for w in weights if id(w) in _DEFERRED]
        bias_only = False
    if not items and not bias_only:
        return
    dev = items[0][0].device if items else next(iter(_DEFERRED_BIAS.values()))[0].device
    _amax_reserve(dev, 8192)
    main = torch.cuda.current_stream()
    side = _side_stream(dev)
    ready = torch.cuda.Event()
    ready.record(main)
    side.wait_event(ready)  # everything enqueued so far (the operands) precedes the side stream's launches
    bias_pending = dict(_DEFERRED_BIAS)
    with torch.cuda.stream(side):
        for weight, its in sorted(items, key=lambda wi: (-wi[0].numel(), wi[0].data_ptr())):
            _wgrad_split_batch(its, weight)
            if _SIDE["on_ready"] is not None and final:
                _READY_DONE.add(id(weight))
                _SIDE["on_ready"](weight)  # (a collective started here orders itself behind the side stream)
        _flush_bias_grads()
        done = torch.cuda.Event()
        done.record(side)
    # the operands were allocated on the main stream: they stay referenced until the main stream has been made to wait
    # for `done` (deferred_wgrad's exit) -- only then may the allocator hand their memory to later main-stream kernels
    _SIDE["keep"].append((items, bias_pending))
    _SIDE["done"] = done


def wgrad_on_side_stream(launch, operands, need_amax=True) -> None:
    """Run one layer's weight-gradient launch(es) on the side stream (inside `deferred_wgrad()` only: its exit is where the
    main stream waits): the data-gradient chain of the backward pass goes on without them.  `operands`: the tensors the
    launch reads -- their operand maxima are taken on the main stream first (a reduction launched on the side stream
    would leave a tag the main stream may consume unordered), and they stay referenced until the main stream has waited."""
    if _DEFERRED is None or not WGRAD_STREAM or not operands[0].is_cuda:
        launch()
        return
    if need_amax:
        for t in operands:
            if t is not None:
                amax_for(t)
    _amax_reserve(operands[0].device, 64)
    main = torch.cuda.current_stream()
    dev = operands[0].device
    side = _side_stream(dev)
    ready = torch.cuda.Event()
    ready.record(main)
    side.wait_event(ready)
    with torch.cuda.stream(side):
        launch()
        done = torch.cuda.Event()
        done.record(side)
    _SIDE["keep"].append(operands)
    _SIDE["done"] = done


_FLUSH_AFTER = None  # deferred_wgrad(flush_after=n | (n1, n2, ..)): a weight's records are launched (side stream) whenever
_FLUSH_SEEN = {}     # the number of steps recorded for it so far (counted here) reaches one of these
_READY_DONE = set()  # weights whose gradient slice was handed to `on_ready` (its reduction may be running): no more records
_VGG_STEPS = False   # ... and the vgg layers' weight gradients are recorded per step too (the encoder / decoder ran per step)
VGG_WGRAD_BATCH = os.environ.get("RAC_VGG_WGRAD_BATCH", "1") == "1"


@contextlib.contextmanager
def deferred_wgrad(on_ready=None, flush_after=None, vgg_steps=False):
    """`on_ready(weight)` is called after each weight's batched launch is enqueued (its gradient is then complete
    in stream order): the trainer starts that slice's data-parallel all-reduce there.
    `flush_after` = n or (n1, n2, ..) (the per-step autograd path of a T-step window: T): whenever a weight has that many
    recorded steps, what is recorded is launched -- time-batched -- on the side stream, under the rest of the backward pass
    (which walks the window from its last step to its first): with T a weight's launch starts the moment its last operand
    exists, instead of behind the whole pass, where 4.8 ms stood exposed at cfg2 with every frame fed back
    (profiles/r06a_*).  Measured at cfg2, every frame fed back, same-day boxes: at exit 1.586 x the teacher-forced step, T
    1.469 x (3.5 ms still exposed: every weight completes inside the last fifth of the pass), (T - 1, T) 1.536 x -- the
    earlier launches take from the latency-bound data-gradient chain more than they hide.  Whatever is left follows when
    the context exits, on the same stream.  The hand-scheduled core launches its chains' gradients itself.
    `vgg_steps`: the encoder and the decoder ran once per step too (a window that feeds predicted frames back): their
    layers' weight gradients are time-batched the same way -- one launch over n steps' pixels instead of n launches of a
    fifth of the rows each (0.06-0.12 of the pipe at 16 images per launch)."""
    global _DEFERRED, _FLUSH_AFTER, _VGG_STEPS
    if not DEFER_WGRAD or _DEFERRED is not None:
        yield
        return
    _DEFERRED = {}
    if flush_after and WGRAD_STREAM and WGRAD_CHAIN_FLUSH:
        pts = (flush_after,) if isinstance(flush_after, int) else tuple(flush_after)
        _FLUSH_AFTER = frozenset(int(v) for v in pts if v and v > 0) or None
    else:
        _FLUSH_AFTER = None
    _FLUSH_SEEN.clear()
    _READY_DONE.clear()
    _VGG_STEPS = bool(vgg_steps) and _FLUSH_AFTER is not None and VGG_WGRAD_BATCH
    _SIDE["on_ready"] = on_ready
    try:
        yield
        if _FLUSH_AFTER is not None and (_DEFERRED or _DEFERRED_BIAS):
            # the rest of what was launched early goes to the SAME stream: a weight's (and a bias's) two launches both
            # read-modify-write its gradient, and only stream order keeps them apart
            flush_deferred_wgrads_early(None)  # (also when only bias column sums are left)
        pending, _DEFERRED = _DEFERRED, None
        # largest first: their all-reduces are the longest and overlap the remaining launches
        # (ties broken by address: every data-parallel rank issues its all-reduces in the same order)
        for weight, items in sorted(pending.values(), key=lambda wi: (-wi[0].numel(), wi[0].data_ptr())):
            _wgrad_split_batch(items, weight)
            if on_ready is not None:
                on_ready(weight)
        _flush_bias_grads()
    finally:
        _DEFERRED = None
        _FLUSH_AFTER = None
        _READY_DONE.clear()
        _VGG_STEPS = False
        _DEFERRED_BIAS.clear()
        if _SIDE["done"] is not None:  # (also on an exception: nothing may outlive the side stream's reads)
            torch.cuda.current_stream().wait_event(_SIDE["done"])
            _SIDE["done"] = None
        _SIDE["keep"].clear()
        _SIDE["on_ready"] = None


def conv_wgrad_split_acc(dy, x0, x1, weight, defer=False):
    """weight.grad += dW on the split-precision pipe.
    `defer`: inside `deferred_wgrad()` only record the operands (the caller must not modify them afterwards)."""
    if defer and _DEFERRED is not None:
        if id(weight) in _READY_DONE:
            raise _lib.RacError("a weight gradient was recorded after the weight's data-parallel reduction had started "
                                "(more steps than deferred_wgrad's flush_after counted on)")
        rec = _DEFERRED.setdefault(id(weight), (weight, []))[1]
        rec.append((dy, x0, x1))
        seen = _FLUSH_SEEN[id(weight)] = _FLUSH_SEEN.get(id(weight), 0) + 1
        if _FLUSH_AFTER is not None and seen in _FLUSH_AFTER:
            for t_ in (dy, x0, x1):  # operand maxima on THIS stream (a reduction launched on the side stream would leave a
                if t_ is not None:   # tag this stream's consumers could read unordered)
                    amax_for(t_)
            for dy_, x0_, x1_ in rec[:-1]:
                for t_ in (dy_, x0_, x1_):
                    if t_ is not None:
                        amax_for(t_)
            # (only the last flush point may start the weight's reduction: earlier ones leave steps to come)
            flush_deferred_wgrads_early([weight], final=seen == max(_FLUSH_AFTER))
        return
    _wgrad_split_batch([(dy, x0, x1)], weight)


def wgrad_split_ok(x0, x1, weight) -> bool:
    """Shapes rac_conv2d_wgrad_split takes (anything else goes to the exact-fp32 MFMA kernel)."""
    co, _, k, _ = weight.shape
    c0 = x0.shape[3]
    c1 = x1.shape[3] if x1 is not None else 0
    return SPLIT_GEMM and k in (3, 5) and co % 16 == 0 and c0 % 8 == 0 and c1 % 8 == 0 and (c1 == 0 or c0 % 64 == 0)


def plan_wgrad_split(tiles: int, groups: int) -> int:
    """K splits of a weight-gradient launch: 512 workgroups run at a time (2 per CU); pick the split whose last round
    is fullest (each extra split costs a slab write + read of the gradient)."""
    return _plan_wgrad_split(tiles, groups, os.environ.get("RAC_WGRAD_SPLITK"))


@functools.lru_cache(maxsize=None)
def _plan_wgrad_split(tiles, groups, forced):
    if forced:
        return max(1, min(int(forced), groups))
    if tiles >= 768:
        # two or more rounds of workgroups without splitting K: a split only evens out the last round, and pays for it
        # with a slab of the whole gradient written, read back and added (the 5x5 gate weights: 210 MB per slab).  Measured
        # (round 4, T = 5 steps of B = 16, g 512, launch + combine): 5x5 (1280 tiles) 1.21 ms unsplit against 1.31 (x 2),
        # 1.40 (x 3), 1.50 (x 4); in the train step 1.03 ms against 0.96 + 0.12.  The 3x3 gate weights (768 tiles = 1.5
        # rounds) kept a x 2 in round 4 (0.43 + 0.03 ms against 0.50 unsplit); with round 5's kernel the launch is short
        # enough that the slab costs more than the uneven last round: 0.526-0.533 ms unsplit against 0.546-0.554 (x 2),
        # 0.575 (x 3) for launch + combine (tools/bench_gemm.py wgrad 16 512 3, T = 5).
        return 1
    best, best_eff = 1, 0.0
    for ns in range(1, min(groups, 64) + 1):
        rounds = tiles * ns / 512.0
        eff = rounds / -(-rounds // 1) - 0.03 * (ns - 1) if rounds > 1 else rounds - 0.002 * (ns - 1)
        if eff > best_eff + 1e-9:
            best, best_eff = ns, eff
    return best


def _wgrad_split_batch(items, weight):
    """weight.grad += sum over the recorded (dy, x0, x1) triples: ONE launch of the column-walking split-precision
    kernel per 16 time steps (operands in their natural NHWC layout; no transposed copies)."""
    dy, x0, x1 = items[0]
    B, H, W, Cout = dy.shape
    Co, Cin, k, _ = weight.shape
    C0 = x0.shape[3]
    if not wgrad_split_ok(x0, x1, weight):
        for dy_t, x0_t, x1_t in items:
            if x1_t is None and x0_t.shape[3] > Cin:
                wgrad_padded_acc(dy_t, x0_t, weight)
            else:
                conv_wgrad_acc(dy_t, x0_t, x1_t, weight)
        return
    ci_real = Cin
    if x1 is None and C0 > Cin:  # x0 carries zero pad channels (32-channel chunks): gradient of the padded weight
        Cin = C0
    sp = stream_ptr()
    dev = dy.device
    fresh = False
    if Cin != ci_real:
        g = torch.empty((Cout, k, k, Cin), device=dev, dtype=torch.float32)  # the first launch writes (accumulate 0)
    else:
        # a stale buffer (lazy zero_grad) is overwritten whole by the first launch: every element of dw is stored by the
        # K split 0 workgroup of its tile, with or without time steps to sum
        fresh = weight.grad is not None and take_stale(weight.grad)
        g = weight_mem(grad_buffer(weight))
    n = Cout * k * k * Cin
    tiles = _cdiv(Cout, 128) * (_cdiv(C0, 64) + _cdiv(Cin - C0, 64)) * k
    keep = []
    for lo in range(0, len(items), _lib.WGRAD_MAX_STEPS):
        chunk = items[lo:lo + _lib.WGRAD_MAX_STEPS]
        # steps whose second source is known to be all zeros (a ConvLSTM's first step) go first: their share of the x1
        # half of the gradient is skipped (the order of the sum over steps is free)
        chunk = sorted(chunk, key=lambda it: 0 if is_zero(it[2]) else 1)
        n_zero = sum(1 for it in chunk if is_zero(it[2]))
        T = len(chunk)
        readers = (_cdiv(C0, 64) + _cdiv(Cin - C0, 64)) * k
        presplit = WGRAD_PRESPLIT and readers >= WGRAD_PRESPLIT_MIN_READERS and Cin == ci_real
        all_ky = WGRAD_ALLKY and k == 3 and Cout <= 128 and H % 32 == 0 and n_zero == 0 and not presplit
        nseg = 1
        if all_ky:
            # tiles are (co 64, ci 64) only: K is split into (32-row group, column segment) units until ~2.5 workgroups
            # sit on every CU -- these layers move few FLOPs per byte, so it is bytes in flight that hide the HBM latency
            tiles_ak = _cdiv(Cout, 64) * (_cdiv(C0, 64) + _cdiv(Cin - C0, 64))
            groups = T * _cdiv(B * H, 32)
            while nseg * 2 <= W // 8 and W % (nseg * 2) == 0 and tiles_ak * groups * nseg < 640:
                nseg *= 2
            forced = os.environ.get("RAC_WGRAD_SPLITK")
            ns = min(groups * nseg, int(forced) if forced else max(1, 1024 // tiles_ak), 1024)
        else:
            ns = plan_wgrad_split(tiles, T * _cdiv(B * H, 32))
        slabs = torch.empty((ns - 1, n), device=dev, dtype=torch.float32) if ns > 1 else None
        a = WgradArgs(B=B, H=H, W=W, ksize=k, Cin=Cin, Cout=Cout, a_split=C0, T=T, nsplit=ns,
                      accumulate=0 if ((Cin != ci_real or fresh) and lo == 0) else 1,
                      dw=ptr(g), slabs=ptr(slabs), slab_stride=n, x1_zero_steps=n_zero, presplit=0,
                      all_ky=1 if all_ky else 0, col_segments=nseg)
        for t, (dy_t, x0_t, x1_t) in enumerate(chunk):
            assert dy_t.shape == dy.shape and dy_t.is_contiguous() and x0_t.is_contiguous()
            a.dy[t], a.x0[t], a.x1[t] = ptr(dy_t), ptr(x0_t), ptr(x1_t)
            a.dy_amax[t], a.x0_amax[t] = ptr(amax_for(dy_t)), ptr(amax_for(x0_t))
            a.x1_amax[t] = ptr(amax_for(x1_t)) if x1_t is not None else None
        # every operand tile is staged by one workgroup per input-channel tile and kernel row: where that is many (the
        # ConvLSTM gate weights: 48 .. 160), split the operands into their fp16 parts ONCE instead of in each of them
        if presplit:
            two = chunk[0][2] is not None

            def split(tensors, slots):
                n_el = tensors[0].numel()
                parts = torch.empty((len(tensors), 2, n_el), device=dev, dtype=torch.float16)
                xs = (C.c_void_p * len(tensors))(*[ptr(t_) for t_ in tensors])
                ps = (C.c_void_p * len(tensors))(*[ptr(parts[i]) for i in range(len(tensors))])
                am = (C.c_void_p * len(slots))(*[ptr(s_) for s_ in slots])
                call("rac_split_steps", xs, ps, len(tensors), n_el, am, len(slots), sp)
                return parts
            x_slots = [amax_for(it[1]) for it in chunk] + ([amax_for(it[2]) for it in chunk] if two else [])
            dy_p = split([it[0] for it in chunk], [amax_for(it[0]) for it in chunk])
            x0_p = split([it[1] for it in chunk], x_slots)
            x1_p = split([it[2] for it in chunk[n_zero:]], x_slots) if two and n_zero < T else None
            for t in range(T):
                a.dy[t], a.x0[t] = ptr(dy_p[t]), ptr(x0_p[t])
                if two:  # (the all-zero steps' x1 is never read: any valid pointer)
                    a.x1[t] = ptr(x1_p[t - n_zero]) if t >= n_zero else ptr(x0_p[t])
            a.presplit = 1
            keep.extend([dy_p, x0_p, x1_p])
        if SHAPE_LOG is not None:
            _log_shape("wgrad16", WGRAD, k, Cout, Cin * k * k, T * B * H * W, 2500.0 / 3)
        call("rac_conv2d_wgrad_split", C.byref(a), sp)
        if ns > 1:
            call("rac_slab_accumulate", ptr(slabs), ns - 1, n, ptr(g), n, sp)
        keep.append(slabs)
    if Cin != ci_real:
        call("rac_unpad_add", ptr(g), Cin, ptr(weight_mem(grad_buffer(weight))), ci_real, Cout * k * k, sp)


# --------------------------------------------------------------------------- #
# zeroed fp64 scratch for the per-channel reductions (BatchNorm statistics and their backward sums)
# --------------------------------------------------------------------------- #
# ~190 tiny zero-fill launches per train step otherwise: `begin_step()` zeroes one arena with one launch and
# `zeros64()` hands out slices of it until the next `begin_step()`.
_ARENA = {"buf": None, "used": 0}
_ARENA_DOUBLES = 1 << 20


def begin_step(device) -> None:
    """Call once per train step before the first kernel: every slice handed out since the last call is recycled."""
    buf = _ARENA["buf"]
    if buf is None or buf.device != torch.device(device):
        buf = _ARENA["buf"] = torch.empty(_ARENA_DOUBLES, device=device, dtype=torch.float64)
    buf.zero_()
    _ARENA["used"] = 0


def zeros64(shape, device) -> torch.Tensor:
    n = 1
    for d in shape:
        n *= d
    buf, used = _ARENA["buf"], _ARENA["used"]
    if buf is None or buf.device != torch.device(device) or used + n > _ARENA_DOUBLES:
        return torch.zeros(shape, device=device, dtype=torch.float64)
    _ARENA["used"] = used + (n + 1) // 2 * 2  # keep slices 16-byte aligned
    return buf[used:used + n].view(shape)


# --------------------------------------------------------------------------- #
# autograd functions
# --------------------------------------------------------------------------- #
class ConvBias(torch.autograd.Function):
    """Conv(k x k, same) + bias over [x0 | x1]; optional sigmoid epilogue.  (nn.Conv2d with bias:
    reference dynamics.py:496-513, lstm.py:273-274.)  If the weight's Cin is not a multiple of 4, x0 must
    carry the zero pad channels (see `padded_weight`)."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, act, frozen=False):
        ci = weight.shape[1]
        # `frozen` = not torch.is_grad_enabled() at the call site (grad mode is always off inside forward())
        ctx.pad32 = (x1 is None and x0.shape[3] == ci + (-ci) % 32 and x0.shape[3] != ci + pad4(ci) and act == ACT_NONE)
        if ctx.pad32:
            # TileCat padded the input to whole 32-channel chunks so that the conv runs split-precision
            y = conv_forward_split(x0, None, padded_weight(weight, x0.shape[3]), bias,
                                   per_image=frozen and per_image_ok(x0.shape[1], x0.shape[2]))
            if not frozen:
                ctx.save_for_backward(x0, None, weight, bias, None)
                ctx.act, ctx.padded, ctx.split = act, False, False
                ctx.amax = (amax_tag(x0), None)
            return y
        padded = ci % 4 != 0 and x1 is None and x0.shape[3] == ci + pad4(ci)
        w = padded_weight(weight) if padded else weight
        B, H, W, _ = x0.shape
        # the split-precision kernels where the shape allows (the NormConvLSTM gate convs)
        ctx.split = (SPLIT_GEMM and act == ACT_NONE and not padded and x0.shape[3] % 32 == 0 and weight.shape[0] >= 128
                     and split_supported(H, W, weight.shape[2], ci, weight.shape[0], x0.shape[3] if x1 is not None else 0))
        if ctx.split and frozen and not per_image_ok(H, W):
            ctx.split = False
        if ctx.split:
            y = conv_forward_split(x0, x1, w, bias, per_image=frozen)
        else:
            y = conv_forward(x0, x1, w, bias, act=act, allow_split=(act == ACT_NONE), frozen=frozen)
        ctx.save_for_backward(x0, x1, weight, bias, y if act != ACT_NONE else None)
        ctx.act, ctx.padded = act, padded
        ctx.amax = (amax_tag(x0), amax_tag(x1))
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, weight, bias, y = ctx.saved_tensors
        retag(x0, ctx.amax[0]), retag(x1, ctx.amax[1])
        dy = dy.contiguous()
        if ctx.act != ACT_NONE:
            d = torch.empty_like(dy)
            call("rac_act_bwd", ptr(dy), ptr(y), ctx.act, ptr(d), dy.numel(), stream_ptr())
            dy = d
        C0 = x0.shape[3]
        C1 = x1.shape[3] if x1 is not None else 0
        dx0 = dx1 = None
        if ctx.pad32:  # both gradients on the split pipe, against the 32-channel-padded weight
            if ctx.needs_input_grad[0]:
                dx0, _ = conv_dgrad_split(dy, padded_weight(weight, C0), C0, 0)
            if weight.requires_grad:
                conv_wgrad_split_acc(dy, x0, None, weight, defer=True)  # unpads into weight.grad
            if bias is not None and bias.requires_grad:
                bias_grad_acc(dy, bias)
            return dx0, None, None, None, None, None
        if ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
            if ctx.split:
                dx0, dx1 = conv_dgrad_split(dy, weight, C0, C1)
            else:
                dx0, dx1 = conv_dgrad(dy, padded_weight(weight) if ctx.padded else weight, C0, C1)
        if weight.requires_grad:
            if ctx.padded:
                wgrad_padded_acc(dy, x0, weight)
            elif ctx.split:
                conv_wgrad_split_acc(dy, x0, x1, weight, defer=True)  # time-batched inside deferred_wgrad()
            else:
                conv_wgrad_acc(dy, x0, x1, weight)
        if bias is not None and bias.requires_grad:
            bias_grad_acc(dy, bias)
        return dx0, dx1, None, None, None, None


def gauss_head_ok(h_shape, weight: torch.Tensor) -> bool:
    """The merged mu | logvar head runs split-precision: channel counts in whole 32-chunks, a supported map size."""
    B, H, W, g = h_shape
    n2, ci, k, _ = weight.shape
    return (SPLIT_GEMM and ci == g and g % 32 == 0 and n2 % 32 == 0 and split_supported(H, W, k, g, n2, 0)
            and per_image_ok(H, W))


class GaussHead(torch.autograd.Function):
    """mu, logvar = Conv(h) twice (lstm.py:273-274) as ONE conv with the two weights stacked along Cout (`weight` /
    `bias` are views over the two adjacent parameters of the flat buffer, their .grad views over the flat gradient):
    one pass over h forward, one data gradient, one weight gradient, all on the split-precision pipe."""

    @staticmethod
    def forward(ctx, h, weight, bias, frozen=False):
        B, H, W, g = h.shape
        n2 = weight.shape[0]
        z, M = n2 // 2, B * H * W
        h = h if h.is_contiguous() else h.contiguous()
        mu = torch.empty((B, H, W, z), device=h.device, dtype=torch.float32)
        lv = torch.empty_like(mu)
        k = weight.shape[2]
        split = 1 if frozen else plan_split_k(M, n2, k * k * _cdiv(g, 32), tile128_only=True)
        if split > 1:
            slabs, split, stride = conv_forward_split(h, None, weight, want_slabs=True)
            call("rac_slab_reduce2", ptr(slabs), split, stride, ptr(bias), ptr(mu), ptr(lv), M, n2, z, None, None,
                 stream_ptr())
        else:
            y = conv_forward_split(h, None, weight, bias, per_image=frozen)
            call("rac_slab_reduce2", ptr(y), 1, M * n2, None, ptr(mu), ptr(lv), M, n2, z, None, None, stream_ptr())
        if not frozen:
            ctx.save_for_backward(h, weight, bias)
            ctx.amax = amax_tag(h)
        return mu, lv

    @staticmethod
    def backward(ctx, dmu, dlv):
        h, weight, bias = ctx.saved_tensors
        retag(h, ctx.amax)
        B, H, W, g = h.shape
        z = weight.shape[0] // 2
        dmu = None if dmu is None else dmu.contiguous()
        dlv = None if dlv is None else dlv.contiguous()
        dy = torch.empty((B, H, W, 2 * z), device=h.device, dtype=torch.float32)
        slot = amax_slot(h.device)
        call("rac_cat2_channels", ptr(dmu), z, ptr(dlv), z, ptr(dy), B * H * W, ptr(slot), stream_ptr())
        tag_amax(dy, slot)
        dh = None
        if ctx.needs_input_grad[0]:
            dh, _ = conv_dgrad_split(dy, weight, g, 0)
        if weight.requires_grad:
            conv_wgrad_split_acc(dy, h, None, weight, defer=True)
            bias_grad_acc(dy, bias)
        return dh, None, None, None


class ConvTHead(torch.autograd.Function):
    """ConvTranspose2d(Cin_w -> Cout_w, 3, 1, 1) + bias + Sigmoid (vgg_64.py:218-220).
    weight is the ConvTranspose parameter (Cin_w, Cout_w, 3, 3); its forward is the DGRAD form."""

    @staticmethod
    def forward(ctx, x, weight, bias, frozen=False):
        B, H, W, Ci = x.shape
        Ciw, Cow, k, _ = weight.shape
        assert Ci == Ciw
        weight_mem(weight)
        y = torch.empty((B, H, W, Cow), device=x.device, dtype=torch.float32)
        if HEAD_MFMA and SPLIT_GEMM and x.is_cuda and (Ciw, Cow, k) == (64, 4, 3) and H % 16 == 0 and W % 16 == 0:
            # the matrix pipe: tap-stacked 1 x 1 conv (weights = A operand, 16 pixels = B) + shifted sum
            wt = _derived(weight, "_rac_head_taps", lambda: weight.detach().permute(2, 3, 0, 1).contiguous())
            x = x if x.is_contiguous() else x.contiguous()
            per_image = frozen and per_image_ok(H, W)
            call("rac_head_fwd_split", ptr(x), ptr(amax_for(x, per_image)), 1 if per_image else 0, ptr(wt), ptr(bias),
                 ptr(y), B, H, W, stream_ptr())
        elif HEAD_DIRECT and x.is_cuda and (Ciw, Cow, k) == (64, 4, 3) and H % 8 == 0 and W % 32 == 0:
            # exact-fp32 FMAs, weights as scalar operands (rac_head_fwd): N = 4 is no shape for the matrix pipe
            wt = _derived(weight, "_rac_head_taps", lambda: weight.detach().permute(2, 3, 0, 1).contiguous())
            x = x if x.is_contiguous() else x.contiguous()
            call("rac_head_fwd", ptr(x), ptr(wt), ptr(bias), ptr(y), B, H, W, stream_ptr())
        elif SPLIT_GEMM and Cow <= 32 and split_supported(H, W, k, Ciw, 32):
            # the transposed-conv forward IS a data gradient: forward conv with the (Cow, Ciw) tap-flipped weight, its
            # Cow rows zero-padded to one 32-column tile; only the Cow real columns are computed into y
            pw, wslot = weight_parts(padded_weight(weight, 32), transposed=True)
            x = x if x.is_contiguous() else x.contiguous()
            per_image = frozen and per_image_ok(H, W)
            _split_launch(x, None, amax_for(x, per_image), None, pw, wslot, y, B=B, H=H, W=W, k=k, Cin=Ciw, Cout=Cow,
                          C0=Ciw, act=ACT_SIGMOID, bias=bias, per_image=per_image)
        else:
            conv_raw(DGRAD, x, None, weight, y, B=B, H=H, W=W, ksize=k, Cin=Cow, Cout=Ciw, act=ACT_SIGMOID, bias=bias)
        ctx.save_for_backward(x, weight, bias, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, y = ctx.saved_tensors
        B, H, W, Cow = y.shape
        Ciw, _, k, _ = weight.shape
        d = torch.empty_like(y)
        call("rac_act_bwd", ptr(dy.contiguous()), ptr(y), ACT_SIGMOID, ptr(d), y.numel(), stream_ptr())
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            # dx[ci] = sum_{tap,co} d[p+tap][co] * w[ci][tap][co]  -> FWD form with "Cout"=Ciw, "Cin"=Cow
            if HEAD_DGRAD and d.is_cuda and (Ciw, Cow, k) == (64, 4, 3) and W % 16 == 0:
                # 4 channels in, 64 out: one pass bound by the 256 B it writes per pixel (rac_head_dgrad), not a GEMM tile
                call("rac_head_dgrad", ptr(d), ptr(weight_mem(weight)), ptr(dx), B, H, W, stream_ptr())
            else:
                conv_raw(FWD, d, None, weight, dx, B=B, H=H, W=W, ksize=k, Cin=Cow, Cout=Ciw, a_split=Cow)
        if weight.requires_grad:
            g = grad_buffer(weight)
            # dw[ci][tap][co] += sum_p x[p][ci] * d[p+tap][co]  -> WGRAD with dy:=x, x:=d
            if Ciw == 64 and Cow <= 8 and k == 3 and H % 16 == 0 and W % 16 == 0:
                thin_wgrad_acc(x, d, Cow, weight)
            else:
                conv_raw(WGRAD, d, None, x, g, B=B, H=H, W=W, ksize=k, Cin=Cow, Cout=Ciw, a_split=Cow, accumulate=1,
                         split_k=0)
        if bias.requires_grad:
            bias_grad_acc(d, bias)
        return dx, None, None, None


# One time step's small vgg layers with BatchNorm in ONE launch each way (rac_bn_small_fwd / _bwd: a workgroup owns a slice of
# channels and all rows -- no atomics, no second launch).  Built, tested (the same numbers to 2e-6) and measured SLOWER: with
# every frame fed back 38.0 ms per step against 33.2 (4 MB layers; 34.6 with 2 MB, 42-53 with 8-16 MB), the deployed model's
# stepped window 28.0 against 25.4 -- 32-64 workgroups walking 1 024+ rows of 8 slabs each are bound by what ONE workgroup
# keeps in flight, and lose more than the two launches and their atomics cost.  Off; RAC_BN_SMALL=1 turns it on.
BN_SMALL = os.environ.get("RAC_BN_SMALL", "0") == "1"
BN_FUSED_APPLY = os.environ.get("RAC_BN_FUSED_APPLY", "1") == "1"  # BatchNorm finalize + affine + LeakyReLU in one launch


def bn_apply_act(raw, stats, gamma, beta, rmean, rvar, n_updates, M, Cc, G):
    """y = LeakyReLU(0.2)(BatchNorm_train(raw)) from the fp64 batch statistics (vgg_64.py:12-14): (y, aff) with aff =
    [scale, shift, mean, invstd] per group for the backward pass; running statistics updated `n_updates` times per group."""
    dev = raw.device
    aff = torch.empty((4, G, Cc), device=dev, dtype=torch.float32)
    y = torch.empty_like(raw)
    slot = amax_slot(dev)
    sp = stream_ptr()
    if BN_FUSED_APPLY and Cc % 4 == 0 and (Cc // 4) & (Cc // 4 - 1) == 0 and Cc <= 1024:
        call("rac_bn_apply_act", ptr(stats), M // G, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), BN_MOMENTUM, BN_EPS,
             n_updates, ptr(raw), ACT_LEAKY, ptr(y), ptr(aff[0]), ptr(aff[1]), ptr(aff[2]), ptr(aff[3]), M, Cc, G, ptr(slot), sp)
    else:
        call("rac_bn_finalize", ptr(stats), M // G, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), BN_MOMENTUM, BN_EPS,
             n_updates, ptr(aff[0]), ptr(aff[1]), ptr(aff[2]), ptr(aff[3]), Cc, G, sp)
        call("rac_affine_act", ptr(raw), ptr(aff[0]), ptr(aff[1]), ACT_LEAKY, ptr(y), M, Cc, G, ptr(slot), sp)
    return tag_amax(y, slot), aff


class VggLayer(torch.autograd.Function):
    """Conv3x3(no bias) -> BatchNorm2d -> LeakyReLU(0.2) over [x0 | x1]  (vgg_64.py:8-18).

    training: batch statistics reduced in fp64 inside the conv epilogue; running stats get
    `n_updates` momentum updates (the reference runs the encoder twice per step on the same frame).
    eval: BatchNorm folded into the conv epilogue (`folded` = (scale, shift))."""

    @staticmethod
    def forward(ctx, x0, x1, weight, gamma, beta, rmean, rvar, training, n_updates, folded, groups=1):
        Cout = weight.shape[0]
        # x0 may carry trailing zero channels (the packed frame: to 16-byte rows, or to one whole 32-channel chunk so
        # that the first layer runs split-precision): the conv then runs against the zero-padded weight
        padded = x1 is None and x0.shape[3] != weight.shape[1]
        ctx.padded = padded
        wfull, weight_used = weight, (padded_weight(weight, x0.shape[3]) if padded else weight)
        weight = weight_used
        if not training:
            scale, shift = folded
            c0 = x0.shape[3]
            if (SPLIT_GEMM and Cout >= SPLIT_MIN_COUT
                    and split_supported(x0.shape[1], x0.shape[2], 3, weight.shape[1], Cout, c0 if x1 is not None else 0)):
                # (no tape through the folded BatchNorm: this is the frozen model, scaled image by image)
                y = conv_forward_split(x0, x1, weight, None, act=ACT_LEAKY, scale=scale, shift=shift,
                                       per_image=per_image_ok(x0.shape[1], x0.shape[2]))
            else:
                y = conv_forward(x0, x1, weight, None, act=ACT_LEAKY, scale=scale, shift=shift)
            ctx.mark_non_differentiable(y)  # frozen-model path (CEM / eval): no backward through folded BatchNorm
            return y
        dev = x0.device
        G = ctx.groups = int(groups)  # time steps batched along B: one BatchNorm call of the reference per group
        stats = zeros64((G, 2, Cout), dev)
        c0 = x0.shape[3]
        ctx.split = (SPLIT_GEMM and Cout >= SPLIT_MIN_COUT_TRAIN
                     and split_supported(x0.shape[1], x0.shape[2], 3, weight.shape[1], Cout, c0 if x1 is not None else 0))
        Mx = x0.shape[0] * x0.shape[1] * x0.shape[2]
        ctx.small = bool(ctx.split and G == 1 and BN_SMALL and _lib.load().rac_bn_small_ok(Mx, Cout))
        if ctx.small:
            # one time step's small layer (a window that feeds its frames back): split-K combine + statistics + affine +
            # LeakyReLU in ONE launch whose workgroups each own a slice of channels (rac_bn_small_fwd)
            slabs, n_slabs, stride = conv_forward_split(x0, x1, weight, want_slabs=True)
            raw = slabs[0] if n_slabs == 1 else torch.empty(tuple(slabs.shape[1:]), device=dev, dtype=torch.float32)
            aff = torch.empty((4, 1, Cout), device=dev, dtype=torch.float32)
            y = torch.empty_like(raw)
            slot = amax_slot(dev)
            call("rac_bn_small_fwd", ptr(slabs), n_slabs, stride, ptr(raw), ptr(y), ptr(gamma), ptr(beta), ptr(rmean),
                 ptr(rvar), BN_MOMENTUM, BN_EPS, n_updates, ptr(aff[0]), ptr(aff[1]), ptr(aff[2]), ptr(aff[3]), Mx, Cout,
                 ACT_LEAKY, ptr(slot), stream_ptr())
            tag_amax(y, slot)
            ctx.save_for_backward(x0, x1, wfull, gamma, beta, raw, aff)
            ctx.amax = (amax_tag(x0), amax_tag(x1))
            return y
        if ctx.split:
            raw = conv_forward_split(x0, x1, weight, None, stats=stats, groups=G)
        else:
            raw = conv_forward(x0, x1, weight, None, stats=stats, groups=G)
        M = raw.numel() // Cout
        assert M % G == 0 and (G == 1 or (M // G) % 128 == 0), (M, G)
        y, aff = bn_apply_act(raw, stats, gamma, beta, rmean, rvar, n_updates, M, Cout, G)
        ctx.save_for_backward(x0, x1, wfull, gamma, beta, raw, aff)
        ctx.amax = (amax_tag(x0), amax_tag(x1))
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, weight, gamma, beta, raw, aff = ctx.saved_tensors
        retag(x0, ctx.amax[0]), retag(x1, ctx.amax[1])
        dy = dy.contiguous()
        Cout = weight.shape[0]
        M = raw.numel() // Cout
        G = ctx.groups
        draw = torch.empty_like(raw)
        want_affine = gamma.requires_grad
        if ctx.small:  # (reduce + apply in one launch: rac_bn_small_bwd)
            call("rac_bn_small_bwd", ptr(dy), ptr(raw), ptr(aff[0]), ptr(aff[1]), ptr(aff[2]), ptr(aff[3]), ptr(draw),
                 ptr(grad_buffer(gamma)) if want_affine else None, ptr(grad_buffer(beta)) if want_affine else None, M, Cout,
                 ptr(tag_amax(draw, amax_slot(dy.device))._rac_amax), stream_ptr())
        else:
            sums = zeros64((G, 2, Cout), dy.device)
            call("rac_bn_bwd_reduce", ptr(dy), ptr(raw), ptr(aff[0]), ptr(aff[1]), ptr(aff[2]), ptr(aff[3]), ptr(sums), M,
                 Cout, G, stream_ptr())
            call("rac_bn_bwd_apply", ptr(dy), ptr(raw), ptr(aff[0]), ptr(aff[1]), ptr(aff[2]), ptr(aff[3]), ptr(sums),
                 ptr(draw), ptr(grad_buffer(gamma)) if want_affine else None,
                 ptr(grad_buffer(beta)) if want_affine else None, M, Cout, G,
                 ptr(tag_amax(draw, amax_slot(dy.device))._rac_amax), stream_ptr())
        C0 = x0.shape[3]
        C1 = x1.shape[3] if x1 is not None else 0
        dx0 = dx1 = None
        if ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
            wuse = padded_weight(weight, C0) if ctx.padded else weight
            if ctx.split:
                dx0, dx1 = conv_dgrad_split(draw, wuse, C0, C1)
            else:
                dx0, dx1 = conv_dgrad(draw, wuse, C0, C1)
        if weight.requires_grad:
            Bq, Hq, Wq, _ = draw.shape
            split, padded = ctx.split, ctx.padded

            def wgrad():
                if (padded and x1 is None and weight.shape[1] <= 8 and Cout == 64 and weight.shape[2] == 3
                        and Hq % 16 == 0 and Wq % 16 == 0):
                    # first encoder layer: 64 x 9 x (3..8) sums over all pixels, straight into the unpadded gradient
                    thin_wgrad_acc(draw, x0, weight.shape[1], weight)
                elif split:
                    conv_wgrad_split_acc(draw, x0, x1, weight)  # un-pads into weight.grad where x0 carries pad channels
                elif padded:
                    wgrad_padded_acc(draw, x0, weight)
                else:
                    conv_wgrad_acc(draw, x0, x1, weight)
            # (the layer's weight gradient leaves the data-gradient chain: side stream, see deferred_wgrad)
            if _VGG_STEPS and split and not padded and wgrad_split_ok(x0, x1, weight):
                conv_wgrad_split_acc(draw, x0, x1, weight, defer=True)  # one launch over the window's steps
            else:
                wgrad_on_side_stream(wgrad, (draw, x0, x1), need_amax=split)
        return dx0, dx1, None, None, None, None, None, None, None, None, None


# the TRAINING path's first encoder layer straight from the frame / mask planes on the matrix pipe (rac_first_layer_fwd_split
# without a folded BatchNorm: raw conv output), instead of packing a 32-channel-padded NHWC tensor and running the
# exact-fp32 implicit GEMM over it (round 3: rac_pack_input 0.063 ms + igemm_fast_kernel 0.146 ms per step at ~0.6 TB/s)
FIRST_TRAIN_MFMA = os.environ.get("RAC_FIRST_TRAIN_MFMA", "1") == "1"


class FirstVggLayer(torch.autograd.Function):
    """c1[0] in training mode: Conv3x3(no bias) over [img * (1 - zero_mask) | mask] -> BatchNorm2d (batch statistics, one set
    per time step group) -> LeakyReLU(0.2)  (dynamics.py:578-582 + vgg_64.py:8-18), from the NCHW planes.  The frames are
    data here (no gradient flows into them: teacher-forced windows); the weight gradient reads a small packed copy of the
    input (3 + Cm channels padded to a multiple of 4), built in backward."""

    @staticmethod
    def forward(ctx, img, zero_mask, mask, weight, gamma, beta, rmean, rvar, n_updates, groups):
        B, _, H, W = img.shape
        Cm = mask.shape[1] if mask is not None else 0
        dev = img.device
        raw = torch.empty((B, H, W, 64), device=dev, dtype=torch.float32)
        sp = stream_ptr()
        call("rac_first_layer_fwd_split", ptr(img), ptr(zero_mask), ptr(mask), Cm, ptr(weight_mem(weight.detach())), None,
             None, ACT_NONE, ptr(raw), None, 0, B, H, W, 64, sp)
        G = int(groups)
        M = B * H * W
        stats = zeros64((G, 2, 64), dev)
        call("rac_col_stats", ptr(raw), ptr(stats), M, 64, G, sp)
        y, aff = bn_apply_act(raw, stats, gamma, beta, rmean, rvar, n_updates, M, 64, G)
        ctx.save_for_backward(img, zero_mask, mask, weight, gamma, beta, raw, aff)
        ctx.groups = G
        return y

    @staticmethod
    def backward(ctx, dy):
        img, zero_mask, mask, weight, gamma, beta, raw, aff = ctx.saved_tensors
        dy = dy.contiguous()
        B, H, W, _ = raw.shape
        M, G = B * H * W, ctx.groups
        sp = stream_ptr()
        sums = zeros64((G, 2, 64), dy.device)
        call("rac_bn_bwd_reduce", ptr(dy), ptr(raw), ptr(aff[0]), ptr(aff[1]), ptr(aff[2]), ptr(aff[3]), ptr(sums), M, 64, G,
             sp)
        draw = torch.empty_like(raw)
        want_affine = gamma.requires_grad
        call("rac_bn_bwd_apply", ptr(dy), ptr(raw), ptr(aff[0]), ptr(aff[1]), ptr(aff[2]), ptr(aff[3]), ptr(sums), ptr(draw),
             ptr(grad_buffer(gamma)) if want_affine else None, ptr(grad_buffer(beta)) if want_affine else None, M, 64, G,
             None, sp)
        if weight.requires_grad:
            Cm = mask.shape[1] if mask is not None else 0
            pad = pad4(3 + Cm)
            x0 = torch.empty((B, H, W, 3 + Cm + pad), device=dy.device, dtype=torch.float32)
            call("rac_pack_input", ptr(img), ptr(zero_mask), ptr(mask), Cm, pad, ptr(x0), B, H * W, sp)
            thin_wgrad_acc(draw, x0, 3 + Cm, weight)
        return (None,) * 10


def first_layer_train_ok(img, mask, weight) -> bool:
    return (FIRST_TRAIN_MFMA and FIRST_MFMA and SPLIT_GEMM and not img.requires_grad and first_layer_ok(img, mask, weight)
            and img.shape[0] * (img.shape[-2] // 16) * (img.shape[-1] // 16) > 0)


def vgg_pool_frozen(x0, weight, scale, shift):
    """(y, MaxPool2d(2)(y)) of a FROZEN vgg layer (vgg_64.py:8-18 + the `mp` of ConvEncoder.forward, :104-129; BatchNorm
    folded): one launch where the conv kernel pools in its epilogue.  None when the layer does not run split-precision."""
    Cout = weight.shape[0]
    if not (SPLIT_GEMM and Cout >= SPLIT_MIN_COUT and x0.shape[3] == weight.shape[1] and x0.shape[1] % 2 == 0
            and x0.shape[2] % 2 == 0 and split_supported(x0.shape[1], x0.shape[2], 3, weight.shape[1], Cout, 0)):
        return None
    return conv_forward_split(x0, None, weight, None, act=ACT_LEAKY, scale=scale, shift=shift,
                              per_image=per_image_ok(x0.shape[1], x0.shape[2]), pool=True)


class MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, H, W, Cc = x.shape
        y = torch.empty((B, H // 2, W // 2, Cc), device=x.device, dtype=torch.float32)
        call("rac_maxpool2_fwd", ptr(x), ptr(y), B, H, W, Cc, stream_ptr())
        ctx.save_for_backward(x)
        return retag(y, amax_tag(x))  # max |y| <= max |x|: the input's slot is a valid bound

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, H, W, Cc = x.shape
        dx = torch.empty_like(x)
        call("rac_maxpool2_bwd", ptr(x), ptr(dy.contiguous()), ptr(dx), B, H, W, Cc, stream_ptr())
        return dx


class Upsample2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, h, w, Cc = x.shape
        y = torch.empty((B, 2 * h, 2 * w, Cc), device=x.device, dtype=torch.float32)
        call("rac_upsample2_fwd", ptr(x), ptr(y), B, h, w, Cc, stream_ptr())
        ctx.shape = (B, h, w, Cc)
        return retag(y, amax_tag(x))

    @staticmethod
    def backward(ctx, dy):
        B, h, w, Cc = ctx.shape
        dx = torch.empty((B, h, w, Cc), device=dy.device, dtype=torch.float32)
        call("rac_upsample2_bwd", ptr(dy.contiguous()), ptr(dx), B, h, w, Cc, stream_ptr())
        return dx


class TileCat(torch.autograd.Function):
    """[tile(v0) | tile(v1) | tile(v2) | m0 | m1 | 0-pad to a multiple of 4] along channels
    (dynamics.py:591-607,634-640).  Gradients flow to the maps only (actions / robot states are data)."""

    @staticmethod
    def forward(ctx, v0, v1, v2, m0, m1, frozen=False):
        B, H, W, c0 = m0.shape
        vs = [v for v in (v0, v1, v2) if v is not None]
        vs += [None] * (3 - len(vs))
        ns = [v.shape[1] if v is not None else 0 for v in vs]
        c1 = m1.shape[3] if m1 is not None else 0
        ct = sum(ns) + c0 + c1
        pad = pad4(ct)
        # whole 32-channel chunks where the consumer conv can then run split-precision: maps that fit a tile, and (round 4)
        # the larger maps of the rows kernels too -- the 16x16 latents of a 128x128 model ran their three input convs on
        # the exact-fp32 pipe (21 ms of a 106 ms cfg5 step: rac_conv2d fwd / dgrad / wgrad at 157 TFLOP/s peak)
        if SPLIT_GEMM and ct >= 128 and c0 % 32 == 0 and split_supported(H, W, 3, ct + (-ct) % 32, 128, 0):
            pad = (-ct) % 32
        out = torch.empty((B, H, W, ct + pad), device=m0.device, dtype=torch.float32)
        slot = amax_slot(m0.device, B if frozen else 1)
        call("rac_tilecat_fwd", ptr(vs[0]), ns[0], ptr(vs[1]), ns[1], ptr(vs[2]), ns[2], ptr(m0), c0, ptr(m1), c1, pad,
             ptr(out), B, H * W, ptr(slot), 1 if frozen else 0, stream_ptr())
        tag_amax(out, slot)
        ctx.meta = (sum(ns), c0, c1)
        return out

    @staticmethod
    def backward(ctx, dout):
        nv, c0, c1 = ctx.meta
        dout = dout.contiguous()
        B, H, W, Ct = dout.shape
        M = B * H * W
        dm0 = dm1 = None
        if ctx.needs_input_grad[3]:
            dm0 = torch.empty((B, H, W, c0), device=dout.device, dtype=torch.float32)
            call("rac_slice_channels", ptr(dout), Ct, nv, c0, ptr(dm0), M, stream_ptr())
        if c1 and ctx.needs_input_grad[4]:
            dm1 = torch.empty((B, H, W, c1), device=dout.device, dtype=torch.float32)
            call("rac_slice_channels", ptr(dout), Ct, nv + c0, c1, ptr(dm1), M, stream_ptr())
        return None, None, None, dm0, dm1, None


def embed_frozen_ok(vs, h, z, weight) -> bool:
    """The frozen model's input conv over cat[tile(v...), h, z] can read h in place (see embed_frozen)."""
    B, H, W, g = h.shape
    nv = sum(v.shape[1] for v in vs if v is not None)
    cz = z.shape[3] if z is not None else 0
    cs = nv + cz + (-(nv + cz)) % 32
    return (SPLIT_GEMM and g % 32 == 0 and weight.shape[1] == nv + g + cz and weight.shape[0] >= 128
            and split_supported(H, W, weight.shape[2], g + cs, weight.shape[0], g) and per_image_ok(H, W))


def embed_frozen(vs, h, z, weight, bias) -> torch.Tensor:
    """Conv(cat[tile(v0), tile(v1), tile(v2), h, z]) + bias (dynamics.py:591-607,634-640) for the frozen model without
    the concatenated tensor: the conv's first source is h itself, the second a small [tile(v) | z | 0-pad] map, against a
    cached copy of the weight whose input channels are reordered to [h | v | z | 0]."""
    B, H, W, g = h.shape
    vs = [v for v in vs if v is not None]
    nv = sum(v.shape[1] for v in vs)
    cz = z.shape[3] if z is not None else 0
    pad = (-(nv + cz)) % 32
    vs3 = vs + [None] * (3 - len(vs))
    small = torch.empty((B, H, W, nv + cz + pad), device=h.device, dtype=torch.float32)
    slot = amax_slot(h.device, B)
    call("rac_tilecat_fwd", ptr(vs3[0]), vs3[0].shape[1] if vs3[0] is not None else 0, ptr(vs3[1]),
         vs3[1].shape[1] if vs3[1] is not None else 0, ptr(vs3[2]), vs3[2].shape[1] if vs3[2] is not None else 0,
         None, 0, ptr(z), cz, pad, ptr(small), B, H * W, ptr(slot), 1, stream_ptr())
    tag_amax(small, slot)

    def build():
        w = weight.detach()
        parts = [w[:, nv:nv + g], w[:, :nv], w[:, nv + g:]]
        if pad:
            parts.append(torch.zeros((w.shape[0], pad, w.shape[2], w.shape[3]), device=w.device, dtype=w.dtype))
        return torch.cat(parts, 1).contiguous(memory_format=torch.channels_last)
    w2 = _derived(weight, f"_rac_reordered_{nv}_{cz}_{pad}", build)
    return conv_forward_split(h, small, w2, bias, per_image=True)


class LstmCell(torch.autograd.Function):
    """ConvLSTMCell (lstm.py:109-149): gates = Conv_k(cat(x, h_prev)) + b; i,f,o = sigmoid; g = tanh;
    c = f*c_prev + i*g; h = o*tanh(c).  The gate GEMM writes split-K slabs that the cell kernel sums."""

    @staticmethod
    def forward(ctx, x, h_prev, c_prev, weight, bias, grad_mode=True):
        B, H, W, g = x.shape
        # `grad_mode` = torch.is_grad_enabled() at the call site: needs_input_grad mirrors requires_grad even
        # under torch.no_grad(), and grad mode is always off inside forward()
        need_bwd = grad_mode and any(ctx.needs_input_grad)
        ctx.split = SPLIT_GEMM and split_supported(H, W, weight.shape[2], 2 * g, 4 * g, g)
        frozen = not grad_mode  # no tape: scales per image, K never split (batch-invariant rollouts)
        if frozen and fused_cell_ok(x, weight):
            return convlstm_cell_frozen(x, h_prev, c_prev, weight, bias)
        if ctx.split and frozen and not per_image_ok(H, W):
            ctx.split = False
        if ctx.split:
            slabs, n_slabs, stride = conv_forward_split(x, h_prev, weight, want_slabs=True, per_image=frozen)
        else:
            slabs, n_slabs, stride = conv_forward(x, h_prev, weight, None, want_slabs=True, frozen=frozen)
        h = torch.empty_like(x)
        c = torch.empty_like(x)
        act = torch.empty((B, H, W, 4 * g), device=x.device, dtype=torch.float32) if need_bwd else None
        call("rac_lstm_cell_fwd", ptr(slabs), n_slabs, stride, ptr(bias), ptr(c_prev), ptr(h), ptr(c), ptr(act),
             B * H * W, g, stream_ptr())
        if need_bwd:
            ctx.save_for_backward(x, h_prev, c_prev, weight, bias, act, c)
            ctx.amax = (amax_tag(x), amax_tag(h_prev))
            ctx.h_zero = is_zero(h_prev)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        x, h_prev, c_prev, weight, bias, act, c = ctx.saved_tensors
        retag(x, ctx.amax[0]), retag(h_prev, ctx.amax[1])
        if ctx.h_zero:  # (saved tensors come back as new objects) the weight gradient skips this step's hidden half
            h_prev._rac_zero = True
        B, H, W, g = x.shape
        M = B * H * W
        dgates = torch.empty_like(act)
        dc_prev = torch.empty_like(c)
        dh = dh.contiguous() if dh is not None else None  # locals: the copies must outlive the launch
        dc = dc.contiguous() if dc is not None else None
        slot = amax_slot(act.device)
        call("rac_lstm_cell_bwd", ptr(dh), ptr(dc), ptr(act), ptr(c_prev), ptr(c), ptr(dgates), ptr(dc_prev), M, g,
             ptr(slot), stream_ptr())
        tag_amax(dgates, slot)
        dx = dh_prev = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            if ctx.split:
                dx, dh_prev = conv_dgrad_split(dgates, weight, g, g, need1=ctx.needs_input_grad[1])
            else:
                dx, dh_prev = conv_dgrad(dgates, weight, g, g)
        if weight.requires_grad:
            if ctx.split:
                conv_wgrad_split_acc(dgates, x, h_prev, weight, defer=True)
            else:
                conv_wgrad_acc(dgates, x, h_prev, weight)
        if bias.requires_grad:
            bias_grad_acc(dgates, bias)
        return dx, dh_prev, dc_prev, None, None, None


# --------------------------------------------------------------------------- #
# the recurrent core of a teacher-forced training window, scheduled by hand
# --------------------------------------------------------------------------- #
# Between the encoder and the decoder a window is T steps of: prior ConvLSTM (2 cells), posterior ConvLSTM (2 cells) +
# mu | logvar head + reparameterisation, frame-predictor input conv over cat[tile(a, r), h_t, z_t] + ConvLSTM (2 cells)
# (dynamics.py:591-641, lstm.py:252-286).  Under autograd each of those was a node: the backward pass paid one split-K
# combine per data gradient (rac_slab_reduce2: dx | dh_prev), one accumulation add per hidden state with two consumers,
# and stack / copy kernels for every per-step slice of a batched tensor (profiles/r03i_train_shapes.md: 33 + 39 + 47
# launches per step).  Here the whole recurrence is ONE autograd node: the forward pass calls the kernels directly and
# keeps what backward needs; the backward pass walks t = T-1 .. 0 and hands every data-gradient conv's RAW K-split slabs
# to their consumers as "gradient sources" (rac_grad_src: slabs read through a column window), so that
#   dh[l, t]  = dx-half of layer l+1's slabs at t  +  dh_prev-half of layer l's slabs at t+1  (+ the head's / decoder's)
# is summed inside rac_lstm_cell_bwd_srcs; no combine pass and no add exists for it.  Weight and bias gradients go
# through the same deferred, time-batched launches as before (conv_wgrad_split_acc(..., defer=True)).
RECURRENT_CORE = os.environ.get("RAC_RECURRENT_CORE", "1") == "1"


def _src(t: torch.Tensor, n_slabs: int, row_stride: int, col_off: int = 0):
    """A gradient source over `t`: [n_slabs][M][row_stride] slabs (or one plain [M][row_stride] map), window at col_off."""
    return (t, n_slabs, t.numel() // n_slabs if n_slabs > 1 else 0, row_stride, col_off)


def _pack_srcs(srcs):
    arr = (GradSrc * max(1, len(srcs)))()
    for i, (t, n, stride, row, col) in enumerate(srcs):
        arr[i] = GradSrc(p=ptr(t), slab_stride=stride, n_slabs=n, row_stride=row, col_off=col, reserved=0)
    return arr


def grad_sum(srcs, out: torch.Tensor, C: int, amax=None):
    """out[M][C] = sum of the gradient sources (+ max |out| folded into the slot `amax`)."""
    M = out.numel() // C
    call("rac_grad_sum", _pack_srcs(srcs), len(srcs), ptr(out), M, C, ptr(amax), stream_ptr())
    return out


def conv_dgrad_slabs(dy, weight, cin: int):
    """Raw K-split slabs [split][M][cin] of the data gradient w.r.t. the first `cin` input channels of `weight`'s conv
    (a channel prefix of the transposed weight is a prefix of its parts); nothing combines them: the consumers read
    them as gradient sources."""
    B, H, W, Cout = dy.shape
    k = weight.shape[2]
    M = B * H * W
    pw, wslot = weight_parts(weight, transposed=True)
    split = plan_split_k(M, cin, k * k * _cdiv(Cout, 32), tile128_only=True)
    slabs = torch.empty((split, M, cin), device=dy.device, dtype=torch.float32)
    _split_launch(dy, None, amax_for(dy), None, pw, wslot, slabs, B=B, H=H, W=W, k=k, Cin=Cout, Cout=cin, C0=Cout,
                  split_k=split, slab_stride=M * cin)
    return slabs, split


def recurrent_core_ok(h_all, g: int, z: int, nv: int, cells, head, frame_conv) -> bool:
    """Configurations the hand-scheduled core takes: plain ConvLSTM cells whose gate convs, the merged posterior head and
    the frame predictor's 32-channel-padded input conv all run on the split-precision pipe (g >= 128 at 64x64 / 128x128);
    anything else keeps the autograd path."""
    if not (RECURRENT_CORE and SPLIT_GEMM and h_all.is_cuda and torch.is_grad_enabled()):
        return False
    _, H, W, gg = h_all.shape
    if gg != g or g % 32 or z % 4 or head is None or not gauss_head_ok((1, H, W, g), head[0]):
        return False
    ct = nv + g + z
    cpad = ct + (-ct) % 32
    if not (ct >= 128 and split_supported(H, W, 3, cpad, 128, 0)):  # TileCat's whole-chunk padding rule
        return False
    if not split_supported(H, W, frame_conv.weight.shape[2], cpad, frame_conv.weight.shape[0], 0):
        return False
    for cell in cells:
        w = cell.gates.weight
        if tuple(w.shape[:2]) != (4 * g, 2 * g) or not split_supported(H, W, w.shape[2], 2 * g, 4 * g, g):
            return False
        if not (w.requires_grad and cell.gates.bias.requires_grad):
            return False
    return frame_conv.weight.requires_grad and head[0].requires_grad


# The three ConvLSTM chains of the recurrent core on their own streams.  Under teacher forcing the prior's chain needs
# nothing of the other two (its inputs are the encoder's latents), the posterior's nothing of the frame predictor's, and the
# frame predictor's step t only z_t of the posterior's step t (dynamics.py:600-629) -- and backward mirrors it (the posterior
# needs dz_t of the frame predictor's step t, the prior nothing).  In one stream every launch of a chain -- a 512-workgroup gate
# GEMM that is exactly one round of the chip, a 10 us cell kernel, a data-gradient combine -- waits for the previous one to
# DRAIN (profiles/r04d_sq_summary.md: matrix pipe 65.6 % busy at M = 1024 against 78.1 % on the same kernel at the planner's
# M = 64000); with the chains in three streams one chain's ramp, epilogue and pointwise kernels run under another's main
# loop.  Same kernels on the same operands: results are bit-equal to the one-stream order (tests/test_gpu_model.py).
# MEASURED (round 5, same box, three interleaved pairs): 23.55 -> 23.42 ms per train step (-0.5 %): concurrent chains share
# the CUs, each launch stretches, and what the chip gains is only the drain / ramp overlap -- the gate GEMM's gap to its
# planner-shape rate is per-WORKGROUP (prologue and epilogue of a K / 4 slab), not per launch.  It also makes every
# per-kernel duration (bench.py's live roofline, rocprof) the duration of a SHARED chip.  Off by default
# (RAC_CHAIN_STREAMS=1 turns it on).
# Memory rule: the side streams wait for the main stream where a region starts and the main stream waits for them where it
# ends; tensors handed from one stream to another inside a region stay referenced until its end (the caching allocator
# reuses a freed block in its OWN stream's order only).
CHAIN_STREAMS = os.environ.get("RAC_CHAIN_STREAMS", "0") == "1"
CORE_LAYER_MAJOR = os.environ.get("RAC_CORE_LAYER_MAJOR", "1") == "1"  # a layer's T launches back to back (RecurrentCore)
CORE_BATCH_THIN = os.environ.get("RAC_CORE_BATCH_THIN", "1") == "1"    # ... and the thin convs between the chains once over all steps
_CHAIN = {"dev": None, "prior": None, "post": None}


class _ChainRegion:
    """Streams of one forward / backward pass of the core: `on("prior" | "post" | "fp")` is a context manager that makes
    the chain's stream current (the frame predictor stays on the caller's stream)."""

    def __init__(self, dev, enabled: bool):
        self.enabled = bool(enabled and CHAIN_STREAMS and torch.device(dev).type == "cuda")
        self.main = torch.cuda.current_stream() if torch.device(dev).type == "cuda" else None
        self.keep = []
        self.streams = {}
        if not self.enabled:
            return
        if _CHAIN["dev"] != torch.device(dev):
            _CHAIN.update(dev=torch.device(dev), prior=torch.cuda.Stream(device=dev), post=torch.cuda.Stream(device=dev))
        self.streams = {"prior": _CHAIN["prior"], "post": _CHAIN["post"]}
        _amax_reserve(dev, 16384)  # no fresh (kernel-zeroed) amax arena inside the region
        start = torch.cuda.Event()
        start.record(self.main)
        for st in self.streams.values():
            st.wait_event(start)

    def on(self, chain: str):
        st = self.streams.get(chain)
        return torch.cuda.stream(st) if st is not None else contextlib.nullcontext()

    def event(self, chain: str):
        """An event at the chain's current position (None when the chains share one stream)."""
        if not self.enabled:
            return None
        ev = torch.cuda.Event()
        ev.record(self.streams.get(chain, self.main))
        return ev

    def wait(self, chain: str, ev) -> None:
        if ev is not None:
            self.streams.get(chain, self.main).wait_event(ev)

    def join(self) -> None:
        for name in self.streams:
            self.main.wait_event(self.event(name))
        self.streams = {}
        self.keep.clear()


class RecurrentCore(torch.autograd.Function):
    """h_pred (T*B), mu, logvar (posterior, T*B), h_prior (T*B) = the T-step recurrence over the encoder's latents.

    forward(h_all, prior_all, post_all, *params): `prior_all` / `post_all` are the prior's / posterior's input convs over all
    T*B latents (run by the caller, batched); `ctx_args` carries the modules and per-step data (not tensors autograd
    tracks).  Parameter gradients are accumulated in place (None is returned for them)."""

    @staticmethod
    def forward(ctx, plan, h_all, prior_all, post_all, *params):
        T, B = plan["T"], plan["B"]
        g, z = plan["g"], plan["z"]
        dev = h_all.device
        _, H, W, _ = h_all.shape
        M = B * H * W
        step = lambda t_: t_.view((T, B) + tuple(t_.shape[1:]))
        h_steps, prior_steps, post_steps = step(h_all), step(prior_all), step(post_all)
        one = amax_one(dev)
        h_pred_all = torch.empty_like(h_all)
        h_prior_all = torch.empty_like(h_all)
        mu_all = torch.empty((T * B, H, W, z), device=dev, dtype=torch.float32)
        lv_all = torch.empty_like(mu_all)
        hp_s, hq_s, mu_s, lv_s = step(h_pred_all), step(h_prior_all), step(mu_all), step(lv_all)
        state = {L: list(plan["init_state"][L]) for L in ("prior", "post", "fp")}
        cells = plan["cells"]  # {L: (cell0, cell1)}
        head_w, head_b = plan["head"]
        fconv = plan["frame_conv"]

        def run_cell(L, l, x, h_out=None):
            cell = cells[L][l].gates
            h_prev, c_prev = state[L][l]
            slabs, n_slabs, stride = conv_forward_split(x, h_prev, cell.weight, want_slabs=True)
            h = h_out if h_out is not None else torch.empty_like(x)
            c = torch.empty_like(x)
            act = torch.empty((B, H, W, 4 * g), device=dev, dtype=torch.float32)
            call("rac_lstm_cell_fwd", ptr(slabs), n_slabs, stride, ptr(cell.bias), ptr(c_prev), ptr(h), ptr(c), ptr(act), M, g,
                 stream_ptr())
            tag_amax(h, one)  # |h| = |o * tanh(c)| < 1
            state[L][l] = (h, c)
            return h, {"x": x, "h_prev": h_prev, "c_prev": c_prev, "act": act, "c": c}

        vs = plan["vs"]  # per step: the tiled vectors (action, robot state(s)) of the frame predictor's input conv
        nv = plan["nv"]
        ct = nv + g + z
        pad = (-ct) % 32
        staged = CORE_LAYER_MAJOR and not (CHAIN_STREAMS and h_all.is_cuda)  # (chains in order: their weights may arrive in order)
        if not staged:
            param_wait()
        fw_box = []

        def fw_pad_now():  # the frame predictor's padded input-conv weight: rebuilt from the parameter, so behind its update
            if not fw_box:
                param_wait(fconv.weight)
                fw_box.append(padded_weight(fconv.weight, ct + pad))
                weight_parts(fw_box[0])
            return fw_box[0]
        if not staged:
            fw_pad_now()
        # the noise of every step, drawn in the reference's order (the prior's draw, dropped, before the posterior's)
        eps_all = []
        for t in range(T):
            if plan["draw_prior_noise"]:
                plan["eps_fn"](mu_s[t])
            eps_all.append(plan["eps_fn"](mu_s[t]))
        # (operand parts of every weight of the core are refreshed -- one launch over ALL weights -- on the caller's stream)
        for L in ("prior", "post", "fp"):
            for cell in cells[L]:
                weight_parts(cell.gates.weight)
        weight_parts(head_w)
        region = _ChainRegion(dev, True)
        tape = [dict() for _ in range(T)]
        h0 = {L: [None] * T for L in ("prior", "post", "fp")}
        z_all, xf_all, z_ready = [None] * T, [None] * T, [None] * T

        # the pieces of one time step, each a closure over t (a layer of a ConvLSTM depends on its own previous step and on
        # the layer BELOW at the same step, never on the layer above)
        def prior_l0(t):
            h0["prior"][t], tape[t]["prior0"] = run_cell("prior", 0, retag(prior_steps[t], amax_tag(prior_all)))

        def prior_l1(t):
            _, tape[t]["prior1"] = run_cell("prior", 1, h0["prior"][t], hq_s[t])

        def post_l0(t):
            h0["post"][t], tape[t]["post0"] = run_cell("post", 0, retag(post_steps[t], amax_tag(post_all)))

        def post_l1(t):
            sp = stream_ptr()
            h_post, tape[t]["post1"] = run_cell("post", 1, h0["post"][t])
            slabs, split, stride = conv_forward_split(h_post, None, head_w, want_slabs=True)
            call("rac_slab_reduce2", ptr(slabs), split, stride, ptr(head_b), ptr(mu_s[t]), ptr(lv_s[t]), M, 2 * z, z,
                 None, None, sp)
            z_t = torch.empty_like(mu_s[t])
            call("rac_reparam_fwd", ptr(mu_s[t]), ptr(lv_s[t]), ptr(eps_all[t]), ptr(z_t), z_t.numel(), sp)
            z_all[t] = z_t
            tape[t].update(h_post=h_post, eps=eps_all[t])

        def fp_in(t):
            sp = stream_ptr()
            v3 = list(vs[t]) + [None] * (3 - len(vs[t]))
            cat = torch.empty((B, H, W, ct + pad), device=dev, dtype=torch.float32)
            slot = amax_slot(dev)
            call("rac_tilecat_fwd", ptr(v3[0]), v3[0].shape[1] if v3[0] is not None else 0, ptr(v3[1]),
                 v3[1].shape[1] if v3[1] is not None else 0, ptr(v3[2]), v3[2].shape[1] if v3[2] is not None else 0,
                 ptr(h_steps[t]), g, ptr(z_all[t]), z, pad, ptr(cat), B, H * W, ptr(slot), 0, sp)
            tag_amax(cat, slot)
            xf_all[t] = conv_forward_split(cat, None, fw_pad_now(), fconv.bias)
            tape[t]["cat"] = cat

        def fp_l0(t):
            h0["fp"][t], tape[t]["fp0"] = run_cell("fp", 0, xf_all[t])

        def fp_l1(t):
            _, tape[t]["fp1"] = run_cell("fp", 1, h0["fp"][t], hp_s[t])

        batch_thin = CORE_LAYER_MAJOR and CORE_BATCH_THIN and not region.enabled and T > 1
        thin = {}

        def post_l1_cell(t):  # (batched form: the posterior's top layer writes slice t of one tensor)
            if t == 0:
                thin["h_post"] = torch.empty((T * B, H, W, g), device=dev, dtype=torch.float32)
            h_post = step(thin["h_post"])[t]
            _, tape[t]["post1"] = run_cell("post", 1, h0["post"][t], h_post)
            tape[t]["h_post"] = h_post

        def post_head_all():
            # mu | logvar head and the reparameterisation of ALL steps in one launch each (M = T B H W rows instead of T launches
            # of B H W: five 20 us launches at 0.07 of the pipe become one at ~0.4)
            sp = stream_ptr()
            h_all_post = tag_amax(thin["h_post"], one)
            slabs, split, stride = conv_forward_split(h_all_post, None, head_w, want_slabs=True)
            call("rac_slab_reduce2", ptr(slabs), split, stride, ptr(head_b), ptr(mu_all), ptr(lv_all), T * M, 2 * z, z,
                 None, None, sp)
            eps_stack = torch.stack(eps_all)
            zs = torch.empty_like(mu_all)
            call("rac_reparam_fwd", ptr(mu_all), ptr(lv_all), ptr(eps_stack), ptr(zs), zs.numel(), sp)
            thin.update(eps=eps_stack, z=zs)
            for t in range(T):
                tape[t]["eps"] = eps_stack[t]

        def fp_in_all():
            # cat[tile(v) | h_t | z_t] and the frame predictor's input conv over all steps at once
            sp = stream_ptr()
            v3 = [torch.cat([vs[t][k] for t in range(T)]) if k < len(vs[0]) else None for k in range(3)]
            cat = torch.empty((T * B, H, W, ct + pad), device=dev, dtype=torch.float32)
            slot = amax_slot(dev)
            call("rac_tilecat_fwd", ptr(v3[0]), v3[0].shape[1] if v3[0] is not None else 0, ptr(v3[1]),
                 v3[1].shape[1] if v3[1] is not None else 0, ptr(v3[2]), v3[2].shape[1] if v3[2] is not None else 0,
                 ptr(h_all), g, ptr(thin["z"]), z, pad, ptr(cat), T * B, H * W, ptr(slot), 0, sp)
            tag_amax(cat, slot)
            xf = conv_forward_split(cat, None, fw_pad_now(), fconv.bias)
            thin.update(cat=cat, xf=xf)
            xs = step(xf)
            for t in range(T):
                xf_all[t] = retag(xs[t], amax_tag(xf))

        try:
            if region.enabled:  # one stream per chain: time-major, the frame predictor's step t behind the posterior's
                for t in range(T):
                    with region.on("prior"):
                        prior_l0(t), prior_l1(t)
                    with region.on("post"):
                        post_l0(t), post_l1(t)
                        z_ready[t] = region.event("post")
                    region.keep.append(z_all[t])  # (allocated in the posterior's stream, read by the frame predictor's)
                    region.wait("fp", z_ready[t])
                    fp_in(t), fp_l0(t), fp_l1(t)
            elif CORE_LAYER_MAJOR:
                # LAYER-major: a layer's T launches back to back.  Its weights (5x5 gate convs: 210 MB of operand parts, 3x3:
                # 75 MB) then meet the Infinity Cache from the second launch on instead of coming from HBM every time (in
                # time-major order 855 MB of other weights pass between two uses); same kernels, same operands, same bits.
                # `batch_thin`: the two thin convs between the chains (posterior head, frame predictor's input conv) once
                # over all steps -- every step's operands exist before the first is needed in this order.
                order = ((prior_l0, prior_l1, post_l0, post_l1_cell, post_head_all, fp_in_all, fp_l0, fp_l1) if batch_thin
                         else (prior_l0, prior_l1, post_l0, post_l1, fp_in, fp_l0, fp_l1))
                for piece in order:
                    # (optim.FusedAdam's late update arrives group by group: prior, posterior, frame predictor + decoder)
                    if piece is prior_l0:
                        param_wait(cells["prior"][0].gates.weight)
                    elif piece is post_l0:
                        param_wait(cells["post"][0].gates.weight)
                    elif piece in (fp_in, fp_in_all):
                        param_wait(cells["fp"][0].gates.weight)
                    if piece in (post_head_all, fp_in_all):
                        piece()
                        continue
                    for t in range(T):
                        piece(t)
            else:
                for t in range(T):
                    for piece in (prior_l0, prior_l1, post_l0, post_l1, fp_in, fp_l0, fp_l1):
                        piece(t)
        finally:
            region.join()
        # (the node keeps what backward reads, not the caller's closures: `eps_fn` is a bound method of the model, and a
        # model -> output tensor -> grad_fn -> plan -> model cycle would keep a dropped model's 4 GB alive until a full GC)
        ctx.plan = {k: plan[k] for k in ("g", "z", "nv", "cells", "head", "frame_conv")}
        ctx.tape = tape
        ctx.thin = thin if batch_thin else None
        ctx.lv_all = lv_all
        ctx.shape = (T, B, H, W)
        plan["final_state"] = state
        return h_pred_all, mu_all, lv_all, h_prior_all

    @staticmethod
    def backward(ctx, d_hpred, d_mu, d_lv, d_hprior):
        plan, tape = ctx.plan, ctx.tape
        T, B, H, W = ctx.shape
        g, z, nv = plan["g"], plan["z"], plan["nv"]
        M = B * H * W
        dev = ctx.lv_all.device
        cells, fconv = plan["cells"], plan["frame_conv"]
        head_w, head_b = plan["head"]
        ct = nv + g + z
        cpad = ct + (-ct) % 32
        fw_pad = padded_weight(fconv.weight, cpad)
        step = lambda t_: None if t_ is None else t_.contiguous().view((T, B) + tuple(t_.shape[1:]))
        d_hpred, d_mu, d_lv, d_hprior = step(d_hpred), step(d_mu), step(d_lv), step(d_hprior)
        lv_s = ctx.lv_all.view((T, B, H, W, z))
        d_h_all = torch.empty((T, B, H, W, g), device=dev, dtype=torch.float32)
        d_prior_all = torch.empty_like(d_h_all)
        d_post_all = torch.empty_like(d_h_all)
        slot_prior, slot_post = amax_slot(dev), amax_slot(dev)
        carry = {L: [None, None] for L in ("prior", "post", "fp")}  # per layer: (dc_prev, dh_prev source) from step t+1

        def cell_bwd(L, l, rec, srcs):
            cell = cells[L][l].gates
            nxt = carry[L][l]
            if nxt is not None and nxt[1] is not None:
                srcs = srcs + [nxt[1]]
            dgates = torch.empty_like(rec["act"])
            dc_prev = torch.empty_like(rec["c"])
            slot = amax_slot(dev)
            call("rac_lstm_cell_bwd_srcs", _pack_srcs(srcs), len(srcs), ptr(nxt[0]) if nxt is not None else None,
                 ptr(rec["act"]), ptr(rec["c_prev"]), ptr(rec["c"]), ptr(dgates), ptr(dc_prev), M, g, ptr(slot), stream_ptr())
            tag_amax(dgates, slot)
            first = is_zero(rec["h_prev"])  # t = 0: nothing flows into the (all-zero, constant) initial state
            cin = g if first else 2 * g
            slabs, n = conv_dgrad_slabs(dgates, cell.weight, cin)
            conv_wgrad_split_acc(dgates, rec["x"], rec["h_prev"], cell.weight, defer=True)
            bias_grad_acc(dgates, cell.bias)
            carry[L][l] = None if first else (dc_prev, _src(slabs, n, cin, g))
            return _src(slabs, n, cin, 0)

        # Three passes over time, one per ConvLSTM chain (the posterior's step t needs only dz_t of the frame predictor's
        # step t; the prior's nothing of the others), each followed by the launch of ITS weight gradients on the side
        # stream: the frame predictor's run under the posterior's and the prior's backward, not only under the encoder's.
        chain_ws = lambda L, extra: [c.gates.weight for c in cells[L]] + extra
        dcats = [None] * T
        for L in ("prior", "post", "fp"):  # (the data gradients' operand parts: refreshed on the caller's stream)
            for cell in cells[L]:
                weight_parts(cell.gates.weight, transposed=True)
        weight_parts(head_w, transposed=True)
        weight_parts(fw_pad, transposed=True)
        region = _ChainRegion(dev, True)
        dz_ready = [None] * T
        top = {L: [None] * T for L in ("prior", "post", "fp")}  # the top layer's data-gradient slabs per step
        dheads = [None] * T

        def fp_b1(t):  # frame predictor, layer 1 (its h feeds the decoder)
            ext = [_src(d_hpred[t], 1, g)] if d_hpred is not None else []
            top["fp"][t] = cell_bwd("fp", 1, tape[t]["fp1"], ext)

        def fp_b0(t):  # ... layer 0, then the input conv over cat[v | h_t | z_t]
            s0 = cell_bwd("fp", 0, tape[t]["fp0"], [top["fp"][t]])
            dy_f = torch.empty((B, H, W, g), device=dev, dtype=torch.float32)
            slot = amax_slot(dev)
            tag_amax(grad_sum([s0], dy_f, g, slot), slot)
            dcat, n_c = conv_dgrad_slabs(dy_f, fw_pad, cpad)
            dz_ready[t] = region.event("fp")  # the posterior's step t may start
            conv_wgrad_split_acc(dy_f, tape[t]["cat"], None, fconv.weight, defer=True)  # un-pads into weight.grad
            bias_grad_acc(dy_f, fconv.bias)
            grad_sum([_src(dcat, n_c, cpad, nv)], d_h_all[t], g)
            dcats[t] = (dcat, n_c)

        def post_b1(t):  # posterior: reparameterisation + KL gradients -> merged head -> layer 1
            rec = tape[t]
            dcat, n_c = dcats[t]
            dy_h = torch.empty((B, H, W, 2 * z), device=dev, dtype=torch.float32)
            slot = amax_slot(dev)
            call("rac_reparam_head_bwd", _pack_srcs([_src(dcat, n_c, cpad, nv + g)]), 1, ptr(lv_s[t]), ptr(rec["eps"]),
                 ptr(d_mu[t]) if d_mu is not None else None, ptr(d_lv[t]) if d_lv is not None else None, ptr(dy_h), M, z,
                 ptr(slot), stream_ptr())
            tag_amax(dy_h, slot)
            dhead, n_h = conv_dgrad_slabs(dy_h, head_w, g)
            conv_wgrad_split_acc(dy_h, rec["h_post"], None, head_w, defer=True)
            bias_grad_acc(dy_h, head_b)
            dheads[t] = dhead
            top["post"][t] = cell_bwd("post", 1, rec["post1"], [_src(dhead, n_h, g, 0)])

        def post_b0(t):  # ... layer 0 -> its input conv's output
            grad_sum([cell_bwd("post", 0, tape[t]["post0"], [top["post"][t]])], d_post_all[t], g, slot_post)

        def prior_b1(t):  # prior (its z is not used on this path: only its hidden state feeds the batched mu_p / logvar_p heads)
            ext = [_src(d_hprior[t], 1, g)] if d_hprior is not None else []
            top["prior"][t] = cell_bwd("prior", 1, tape[t]["prior1"], ext)

        def prior_b0(t):
            grad_sum([cell_bwd("prior", 0, tape[t]["prior0"], [top["prior"][t]])], d_prior_all[t], g, slot_prior)

        steps = range(T - 1, -1, -1)
        layer_major = CORE_LAYER_MAJOR and not region.enabled
        thin = ctx.thin  # forward ran the posterior head and the frame predictor's input conv over all steps at once
        if thin is not None:
            dy_f_all = torch.empty((T * B, H, W, g), device=dev, dtype=torch.float32)
            slot_f = amax_slot(dev)

            def fp_b0(t):  # layer 0 of the frame predictor: the gradient of its input, slice t of one tensor  # noqa: F811
                s0 = cell_bwd("fp", 0, tape[t]["fp0"], [top["fp"][t]])
                grad_sum([s0], dy_f_all.view(T, B, H, W, g)[t], g, slot_f)

            def fp_in_b_all():  # the input conv's data gradient, its weight gradient and d h, over all steps
                tag_amax(dy_f_all, slot_f)
                dcat, n_c = conv_dgrad_slabs(dy_f_all, fw_pad, cpad)
                conv_wgrad_split_acc(dy_f_all, thin["cat"], None, fconv.weight, defer=True)
                bias_grad_acc(dy_f_all, fconv.bias)
                grad_sum([_src(dcat, n_c, cpad, nv)], d_h_all, g)
                thin["dcat"] = (dcat, n_c)

            def post_head_b_all():  # reparameterisation + KL gradients and the merged head's data gradient, all steps
                dcat, n_c = thin["dcat"]
                dy_h = torch.empty((T * B, H, W, 2 * z), device=dev, dtype=torch.float32)
                slot = amax_slot(dev)
                call("rac_reparam_head_bwd", _pack_srcs([_src(dcat, n_c, cpad, nv + g)]), 1, ptr(ctx.lv_all), ptr(thin["eps"]),
                     ptr(d_mu) if d_mu is not None else None, ptr(d_lv) if d_lv is not None else None, ptr(dy_h), T * M, z,
                     ptr(slot), stream_ptr())
                tag_amax(dy_h, slot)
                dhead, n_h = conv_dgrad_slabs(dy_h, head_w, g)
                conv_wgrad_split_acc(dy_h, tag_amax(thin["h_post"], amax_one(dev)), None, head_w, defer=True)
                bias_grad_acc(dy_h, head_b)
                thin["dhead"] = (dhead, n_h)

            def post_b1(t):  # noqa: F811  (its head gradient: rows t of the batched slabs)
                dhead, n_h = thin["dhead"]
                rows = dhead.view(n_h, T, M, g)[0, t]  # (first slab's rows of step t: the others follow at the slab stride)
                src = (rows, n_h, dhead.numel() // n_h if n_h > 1 else 0, g, 0)
                top["post"][t] = cell_bwd("post", 1, tape[t]["post1"], [src])
        try:
            if layer_major:  # (see forward: a layer's T data-gradient launches back to back)
                for piece in (fp_b1, fp_b0):
                    for t in steps:
                        piece(t)
                if thin is not None:
                    fp_in_b_all()
            else:
                for t in steps:
                    fp_b1(t), fp_b0(t)
            flush_deferred_wgrads_early(chain_ws("fp", [fconv.weight]))
            with region.on("post"):
                if layer_major:
                    if thin is not None:
                        post_head_b_all()
                    for piece in (post_b1, post_b0):
                        for t in steps:
                            piece(t)
                else:
                    for t in steps:
                        region.wait("post", dz_ready[t])
                        post_b1(t), post_b0(t)
                flush_deferred_wgrads_early(chain_ws("post", [head_w]))  # (ordered behind the posterior's stream)
            with region.on("prior"):
                if layer_major:
                    for piece in (prior_b1, prior_b0):
                        for t in steps:
                            piece(t)
                else:
                    for t in steps:
                        prior_b1(t), prior_b0(t)
                if region.enabled:
                    flush_deferred_wgrads_early(chain_ws("prior", []))
        finally:
            region.join()
        # (the tape and the slabs handed between the chains are released only now: behind the join)
        dcats = tape = top = dheads = None
        # every operand of the core's weight gradients exists now: they start on the side stream, under the encoder's backward
        # (and of whatever else was recorded before: the prior's heads)
        flush_deferred_wgrads_early(None)
        ctx.tape = ctx.plan = ctx.thin = None
        flat = lambda t_: t_.view((T * B, H, W, g))
        d_prior_all, d_post_all = tag_amax(flat(d_prior_all), slot_prior), tag_amax(flat(d_post_all), slot_post)
        return (None, flat(d_h_all), d_prior_all, d_post_all) + (None,) * (len(ctx.needs_input_grad) - 4)


NORM_CELL_FUSED = os.environ.get("RAC_NORM_CELL_FUSED", "1") == "1"  # the frozen NormConvLSTMCell's pointwise part in one launch


def norm_cell_frozen_ok(g: int) -> bool:
    return NORM_CELL_FUSED and g % 16 == 0 and (g // 16) & (g // 16 - 1) == 0 and g <= 4096


def norm_cell_frozen(g_ih, g_hh, c_prev, gn_ih, gn_hh, gn_c):
    """(h, c) of a NormConvLSTMCell behind its two gate convs, without a tape (lstm.py:174-198): both GroupNorm(16, 4g),
    the gate activations, the cell update, GroupNorm(16, g) of the cell and h = o tanh(c) in ONE launch
    (rac_norm_lstm_cell_fwd) instead of GroupNorm x 3 + cell + output kernels.  `gn_*` = (weight, bias) of the norms."""
    return _norm_cell_launch(g_ih, g_hh, c_prev, gn_ih, gn_hh, gn_c, False)[:2]


def _norm_cell_launch(g_ih, g_hh, c_prev, gn_ih, gn_hh, gn_c, training, h_out=None):
    B, H, W, g4 = g_ih.shape
    g = g4 // 4
    g_ih = g_ih if g_ih.is_contiguous() else g_ih.contiguous()
    g_hh = g_hh if g_hh.is_contiguous() else g_hh.contiguous()
    c_prev = c_prev if c_prev.is_contiguous() else c_prev.contiguous()
    dev = g_ih.device
    h = h_out if h_out is not None else torch.empty((B, H, W, g), device=dev, dtype=torch.float32)
    c = torch.empty_like(h)
    act = c_raw = stats = None
    if training:
        act = torch.empty((B, H, W, g4), device=dev, dtype=torch.float32)
        c_raw = torch.empty_like(h)
        stats = torch.empty((3, 2, B, 16), device=dev, dtype=torch.float32)  # [ih | hh | c][mean | rstd][image][group]
    call("rac_norm_lstm_cell_fwd", ptr(g_ih), ptr(g_hh), ptr(c_prev), ptr(gn_ih[0]), ptr(gn_ih[1]), ptr(gn_hh[0]),
         ptr(gn_hh[1]), ptr(gn_c[0]), ptr(gn_c[1]), ptr(h), ptr(c), ptr(act), ptr(c_raw),
         ptr(stats[0]) if training else None, ptr(stats[1]) if training else None, ptr(stats[2]) if training else None,
         B, H * W, g, 1e-5, stream_ptr())
    return h, c, act, c_raw, stats, g_ih, g_hh, c_prev


NORM_CELL_BWD_FUSED = os.environ.get("RAC_NORM_CELL_BWD_FUSED", "1") == "1"  # ... and its pointwise backward as ONE launch
NORM_CELL_NODE = os.environ.get("RAC_NORM_CELL_NODE", "1") == "1"  # training: the whole NormConvLSTMCell as ONE autograd node


def norm_cell_node_ok(x, w_ih, w_hh) -> bool:
    B, H, W, g = x.shape
    k = w_ih.shape[2]
    return (NORM_CELL_NODE and x.is_cuda and norm_cell_frozen_ok(g) and SPLIT_GEMM and g % 32 == 0
            and tuple(w_ih.shape[:2]) == (4 * g, g) and tuple(w_hh.shape) == tuple(w_ih.shape)
            and split_supported(H, W, k, g, 4 * g, 0))


class NormLstmCell(torch.autograd.Function):
    """NormConvLSTMCell (lstm.py:151-198) as ONE autograd node: the two gate convs (split-precision pipe), GroupNorm(16, 4g)
    of each, the gate activations, the cell update, GroupNorm(16, g) of the cell and h = o tanh(c).  Forward: two conv
    launches + rac_norm_lstm_cell_fwd; backward: rac_lstm_out_bwd, rac_groupnorm_bwd (cell), rac_lstm_core_bwd,
    rac_groupnorm_bwd x 2, the two data gradients, the deferred (time-batched) weight gradients.  The same kernels as the
    seven-node form (ConvBias x 2, GroupNorm x 3, NormCellCore, LstmOut) minus the bookkeeping between them: at the deployed
    model's size (g 256, 6x8 maps, batch 16) the HOST, not the GPU, bounded the train step (tools/host_breakdown.py)."""

    @staticmethod
    def forward(ctx, x, h_prev, c_prev, w_ih, b_ih, gam_ih, bet_ih, w_hh, b_hh, gam_hh, bet_hh, gam_c, bet_c):
        g_ih = conv_forward_split(x, None, w_ih, b_ih)
        if is_zero(h_prev):  # a window's first step: the conv of the all-zero state is its bias, to the bit
            g_hh = b_hh.detach().view(1, 1, 1, -1).expand(tuple(x.shape[:3]) + (b_hh.numel(),)).contiguous()
        else:
            g_hh = conv_forward_split(h_prev, None, w_hh, b_hh)
        h, c, act, c_raw, stats, g_ih, g_hh, c_prev_c = _norm_cell_launch(g_ih, g_hh, c_prev, (gam_ih, bet_ih), (gam_hh, bet_hh),
                                                                           (gam_c, bet_c), True)
        ctx.save_for_backward(x, h_prev, c_prev_c, w_ih, b_ih, gam_ih, bet_ih, w_hh, b_hh, gam_hh, bet_hh, gam_c, bet_c,
                              g_ih, g_hh, act, c_raw, c, stats)
        ctx.amax = (amax_tag(x), amax_tag(h_prev))
        ctx.h_zero = is_zero(h_prev)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        (x, h_prev, c_prev, w_ih, b_ih, gam_ih, bet_ih, w_hh, b_hh, gam_hh, bet_hh, gam_c, bet_c, g_ih, g_hh, act, c_raw, c,
         stats) = ctx.saved_tensors
        retag(x, ctx.amax[0]), retag(h_prev, ctx.amax[1])
        B, H, W, g = x.shape
        M, HW = B * H * W, H * W
        dev = x.device
        sp = stream_ptr()
        want = gam_ih.requires_grad
        dg_ih, dg_hh = torch.empty_like(act), torch.empty_like(act)
        dc_prev = torch.empty_like(c)
        if NORM_CELL_BWD_FUSED:
            # everything between the incoming gradients and the two convs' output gradients: one launch
            sa, sb = amax_slot(dev), amax_slot(dev)
            dh_c = dh.contiguous() if dh is not None else None  # (locals: the copies must outlive the launch)
            dc_c = dc.contiguous() if dc is not None else None
            call("rac_norm_lstm_cell_bwd", ptr(dh_c), ptr(dc_c), ptr(act), ptr(c), ptr(c_raw), ptr(c_prev), ptr(g_ih),
                 ptr(g_hh), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(gam_ih), ptr(gam_hh), ptr(gam_c), ptr(dg_ih),
                 ptr(dg_hh), ptr(dc_prev), ptr(grad_buffer(gam_ih)) if want else None,
                 ptr(grad_buffer(bet_ih)) if want else None, ptr(grad_buffer(gam_hh)) if want else None,
                 ptr(grad_buffer(bet_hh)) if want else None, ptr(grad_buffer(gam_c)) if want else None,
                 ptr(grad_buffer(bet_c)) if want else None, ptr(sa), ptr(sb), B, HW, g, sp)
            tag_amax(dg_ih, sa), tag_amax(dg_hh, sb)
        else:
            # h = o * tanh(c): gradient on the o slot of the activations and on the normalised cell (+ what the next step sent)
            d_act = dc_n = None
            if dh is not None:
                d_act = torch.empty_like(act)
                dc_n = torch.empty_like(c)
                call("rac_lstm_out_bwd", ptr(dh.contiguous()), ptr(act), ptr(c), ptr(d_act), ptr(dc_n), M, g, sp)
            if dc is not None:
                dc_n = dc.contiguous() if dc_n is None else dc_n.add_(dc)
            dc_raw = None
            if dc_n is not None:
                dc_raw = torch.empty_like(c)
                call("rac_groupnorm_bwd", ptr(dc_n), ptr(c_raw), ptr(gam_c), ptr(stats[2, 0]), ptr(stats[2, 1]), ptr(dc_raw),
                     ptr(grad_buffer(gam_c)) if want else None, ptr(grad_buffer(bet_c)) if want else None, B, HW, g, 16, sp)
            dgates = torch.empty_like(act)
            call("rac_lstm_core_bwd", ptr(dc_raw), ptr(d_act), ptr(act), ptr(c_prev), ptr(dgates), ptr(dc_prev), M, g, sp)
            call("rac_groupnorm_bwd", ptr(dgates), ptr(g_ih), ptr(gam_ih), ptr(stats[0, 0]), ptr(stats[0, 1]), ptr(dg_ih),
                 ptr(grad_buffer(gam_ih)) if want else None, ptr(grad_buffer(bet_ih)) if want else None, B, HW, 4 * g, 16, sp)
            call("rac_groupnorm_bwd", ptr(dgates), ptr(g_hh), ptr(gam_hh), ptr(stats[1, 0]), ptr(stats[1, 1]), ptr(dg_hh),
                 ptr(grad_buffer(gam_hh)) if want else None, ptr(grad_buffer(bet_hh)) if want else None, B, HW, 4 * g, 16, sp)
        dx = dh_prev = None
        if ctx.needs_input_grad[0]:
            dx, _ = conv_dgrad_split(dg_ih, w_ih, g, 0)
        if ctx.needs_input_grad[1] and not ctx.h_zero:
            dh_prev, _ = conv_dgrad_split(dg_hh, w_hh, g, 0)
        if w_ih.requires_grad:
            conv_wgrad_split_acc(dg_ih, x, None, w_ih, defer=True)
            bias_grad_acc(dg_ih, b_ih)
        if w_hh.requires_grad:
            if not ctx.h_zero:  # (an all-zero hidden state: the weight's gradient gets nothing from this step)
                conv_wgrad_split_acc(dg_hh, h_prev, None, w_hh, defer=True)
            bias_grad_acc(dg_hh, b_hh)
        return (dx, dh_prev, dc_prev) + (None,) * 10


# --------------------------------------------------------------------------- #
# the recurrent core for NormConvLSTMCell models (--lstm_group_norm True)
# --------------------------------------------------------------------------- #
# The same idea as RecurrentCore -- the T-step recurrence of a teacher-forced window as ONE autograd node, layer-major,
# gradient maps handed over as sources -- for the cell of the deployed checkpoints (lstm.py:151-198).  Its two gate convs
# are separate (each is normalised on its own), and only conv_hh reads the recurrence: in layer-major order the INPUT half
# of a layer's gate convs runs ONCE over all T steps (M = T B H W rows instead of T launches of B H W: 3 840 instead of 768
# at the deployed model's size, where the per-step GEMMs ran at 0.12-0.31 of the pipe), and so do its data gradient and its
# weight gradient (one record of the batched operands).  Per step: conv_hh, rac_norm_lstm_cell_fwd / _bwd, conv_hh's data
# gradient.
NORM_RECURRENT_CORE = os.environ.get("RAC_NORM_RECURRENT_CORE", "1") == "1"


def norm_recurrent_core_ok(h_all, g: int, z: int, nv: int, cells, head, frame_conv) -> bool:
    if not (NORM_RECURRENT_CORE and RECURRENT_CORE and SPLIT_GEMM and h_all.is_cuda and torch.is_grad_enabled()):
        return False
    _, H, W, gg = h_all.shape
    if gg != g or g % 32 or z % 4 or head is None or not gauss_head_ok((1, H, W, g), head[0]):
        return False
    ct = nv + g + z
    cpad = ct + (-ct) % 32
    if not (ct >= 128 and split_supported(H, W, 3, cpad, 128, 0)):
        return False
    if not split_supported(H, W, frame_conv.weight.shape[2], cpad, frame_conv.weight.shape[0], 0):
        return False
    x_like = h_all[:1]
    for cell in cells:
        ci, ch = cell.ih_gates[0], cell.hh_gates[0]
        if not norm_cell_node_ok(x_like, ci.weight, ch.weight) or not NORM_CELL_BWD_FUSED:
            return False
        if not all(p.requires_grad for p in cell.parameters()):
            return False
    return frame_conv.weight.requires_grad and head[0].requires_grad


def _plain_map(src, g):
    """The tensor behind a gradient source that is one plain [M][g] map (no slabs, no column window), else None."""
    t, n, _, row, col = src
    return t if (n == 1 and row == g and col == 0 and t.is_contiguous()) else None


class NormRecurrentCore(torch.autograd.Function):
    """h_pred (T*B), mu, logvar (posterior, T*B), h_prior (T*B) = the T-step recurrence over the encoder's latents for a
    model of NormConvLSTMCells (same contract as RecurrentCore; parameter gradients accumulate in place)."""

    @staticmethod
    def forward(ctx, plan, h_all, prior_all, post_all, *params):
        T, B, g, z = plan["T"], plan["B"], plan["g"], plan["z"]
        dev = h_all.device
        _, H, W, _ = h_all.shape
        M = B * H * W
        step = lambda t_: t_.view((T, B) + tuple(t_.shape[1:]))
        one = amax_one(dev)
        cells = plan["cells"]
        head_w, head_b = plan["head"]
        fconv = plan["frame_conv"]
        nv = plan["nv"]
        ct = nv + g + z
        pad = (-ct) % 32
        param_wait()
        fw_pad = padded_weight(fconv.weight, ct + pad)
        mu_all = torch.empty((T * B, H, W, z), device=dev, dtype=torch.float32)
        lv_all = torch.empty_like(mu_all)
        eps_all = []
        for t in range(T):  # the noise of every step in the reference's order (the prior's draw, dropped, first)
            if plan["draw_prior_noise"]:
                plan["eps_fn"](step(mu_all)[t])
            eps_all.append(plan["eps_fn"](step(mu_all)[t]))
        for L in ("prior", "post", "fp"):
            for cell in cells[L]:
                weight_parts(cell.ih_gates[0].weight), weight_parts(cell.hh_gates[0].weight)
        weight_parts(head_w), weight_parts(fw_pad)
        state = {L: list(plan["init_state"][L]) for L in ("prior", "post", "fp")}
        tape = {}

        def run_layer(L, l, x_all):
            """All T steps of one layer: the input half of its gate convs once over the window, then step by step."""
            cell = cells[L][l]
            ci, ch = cell.ih_gates[0], cell.hh_gates[0]
            gn = [(m.weight, m.bias) for m in (cell.ih_gates[1], cell.hh_gates[1], cell.c_norm)]
            g_ih_all = conv_forward_split(x_all, None, ci.weight, ci.bias)
            h_all_l = torch.empty((T * B, H, W, g), device=dev, dtype=torch.float32)
            h_s, gi_s = step(h_all_l), step(g_ih_all)
            h_prev, c_prev = state[L][l]
            recs = []
            for t in range(T):
                if is_zero(h_prev):  # the conv of the all-zero initial state is its bias, to the bit
                    g_hh = ch.bias.detach().view(1, 1, 1, -1).expand(B, H, W, 4 * g).contiguous()
                else:
                    g_hh = conv_forward_split(h_prev, None, ch.weight, ch.bias)
                h, c, act, c_raw, stats, _, g_hh, c_prev_c = _norm_cell_launch(gi_s[t], g_hh, c_prev, gn[0], gn[1], gn[2], True,
                                                                               h_out=h_s[t])
                tag_amax(h, one)  # |h| = |o * tanh(c)| < 1
                recs.append({"h_prev": h_prev, "c_prev": c_prev_c, "g_hh": g_hh, "act": act, "c_raw": c_raw, "c": c,
                             "stats": stats})
                h_prev, c_prev = h, c
            state[L][l] = (h_prev, c_prev)
            tape[(L, l)] = {"x_all": x_all, "g_ih_all": g_ih_all, "h_all": h_all_l, "steps": recs}
            return tag_amax(h_all_l, one)

        h_prior_all = run_layer("prior", 1, run_layer("prior", 0, prior_all))
        h_post_all = run_layer("post", 1, run_layer("post", 0, post_all))
        # posterior head + reparameterisation, then the frame predictor's input conv, once over all steps
        sp = stream_ptr()
        slabs, split, stride = conv_forward_split(h_post_all, None, head_w, want_slabs=True)
        call("rac_slab_reduce2", ptr(slabs), split, stride, ptr(head_b), ptr(mu_all), ptr(lv_all), T * M, 2 * z, z, None, None, sp)
        eps_stack = torch.stack(eps_all)
        zs = torch.empty_like(mu_all)
        call("rac_reparam_fwd", ptr(mu_all), ptr(lv_all), ptr(eps_stack), ptr(zs), zs.numel(), sp)
        vs = plan["vs"]
        v3 = [torch.cat([vs[t][k] for t in range(T)]) if k < len(vs[0]) else None for k in range(3)]
        cat = torch.empty((T * B, H, W, ct + pad), device=dev, dtype=torch.float32)
        slot = amax_slot(dev)
        call("rac_tilecat_fwd", ptr(v3[0]), v3[0].shape[1] if v3[0] is not None else 0, ptr(v3[1]),
             v3[1].shape[1] if v3[1] is not None else 0, ptr(v3[2]), v3[2].shape[1] if v3[2] is not None else 0,
             ptr(h_all), g, ptr(zs), z, pad, ptr(cat), T * B, H * W, ptr(slot), 0, sp)
        tag_amax(cat, slot)
        xf_all = conv_forward_split(cat, None, fw_pad, fconv.bias)
        h_pred_all = run_layer("fp", 1, run_layer("fp", 0, xf_all))
        ctx.plan = {k: plan[k] for k in ("g", "z", "nv", "cells", "head", "frame_conv")}
        ctx.tape = tape
        ctx.thin = {"h_post": h_post_all, "eps": eps_stack, "cat": cat}
        ctx.lv_all = lv_all
        ctx.shape = (T, B, H, W)
        plan["final_state"] = state
        return h_pred_all, mu_all, lv_all, h_prior_all

    @staticmethod
    def backward(ctx, d_hpred, d_mu, d_lv, d_hprior):
        plan, tape, thin = ctx.plan, ctx.tape, ctx.thin
        T, B, H, W = ctx.shape
        g, z, nv = plan["g"], plan["z"], plan["nv"]
        M, HW = B * H * W, H * W
        dev = ctx.lv_all.device
        cells, fconv = plan["cells"], plan["frame_conv"]
        head_w, head_b = plan["head"]
        ct = nv + g + z
        cpad = ct + (-ct) % 32
        fw_pad = padded_weight(fconv.weight, cpad)
        cont = lambda t_: None if t_ is None else t_.contiguous()
        d_hpred, d_mu, d_lv, d_hprior = cont(d_hpred), cont(d_mu), cont(d_lv), cont(d_hprior)
        for L in ("prior", "post", "fp"):
            for cell in cells[L]:
                weight_parts(cell.ih_gates[0].weight, transposed=True), weight_parts(cell.hh_gates[0].weight, transposed=True)
        weight_parts(head_w, transposed=True), weight_parts(fw_pad, transposed=True)
        sp = stream_ptr()

        def window(slabs, n, cin, t, col=0):  # rows of step t of batched slabs [n][T M][cin] as a gradient source
            rows = slabs.view(n, T, M, cin)[0, t]
            return (rows, n, slabs.numel() // n if n > 1 else 0, cin, col)

        def layer_bwd(L, l, ext):
            """Backward of all T steps of one layer; `ext(t)` = the sources of dh[t] from outside the layer's own recurrence.
            Returns the batched slabs of the gradient w.r.t. the layer's input (n_slabs given) for whoever produced it."""
            cell = cells[L][l]
            ci, ch = cell.ih_gates[0], cell.hh_gates[0]
            n_ih, n_hh, n_c = cell.ih_gates[1], cell.hh_gates[1], cell.c_norm
            rec = tape[(L, l)]
            dg_ih_all = torch.empty_like(rec["g_ih_all"])
            dgi_s, gi_s = dg_ih_all.view((T, B) + tuple(dg_ih_all.shape[1:])), rec["g_ih_all"].view((T, B) + tuple(dg_ih_all.shape[1:]))
            slot_ih = amax_slot(dev)  # (every step folds its maximum in: the batched tensor's)
            want = n_ih.weight.requires_grad
            gb = lambda p_: ptr(grad_buffer(p_)) if want else None
            dc_next, dh_src = None, None
            for t in range(T - 1, -1, -1):
                r = rec["steps"][t]
                srcs = list(ext(t)) + ([dh_src] if dh_src is not None else [])
                dh = None
                if len(srcs) == 1 and _plain_map(srcs[0], g) is not None:
                    dh = _plain_map(srcs[0], g)
                elif srcs:
                    dh = grad_sum(srcs, torch.empty((B, H, W, g), device=dev, dtype=torch.float32), g)
                dg_hh = torch.empty_like(r["act"])
                dc_prev = torch.empty_like(r["c"])
                slot_hh = amax_slot(dev)
                st = r["stats"]
                call("rac_norm_lstm_cell_bwd", ptr(dh), ptr(dc_next), ptr(r["act"]), ptr(r["c"]), ptr(r["c_raw"]),
                     ptr(r["c_prev"]), ptr(gi_s[t]), ptr(r["g_hh"]), ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(n_ih.weight),
                     ptr(n_hh.weight), ptr(n_c.weight), ptr(dgi_s[t]), ptr(dg_hh), ptr(dc_prev), gb(n_ih.weight), gb(n_ih.bias),
                     gb(n_hh.weight), gb(n_hh.bias), gb(n_c.weight), gb(n_c.bias), ptr(slot_ih), ptr(slot_hh), B, HW, g, sp)
                tag_amax(dg_hh, slot_hh)
                first = is_zero(r["h_prev"])  # nothing flows into the (all-zero, constant) initial state
                if not first:
                    slabs, n = conv_dgrad_slabs(dg_hh, ch.weight, g)
                    dh_src = _src(slabs, n, g, 0)
                    conv_wgrad_split_acc(dg_hh, r["h_prev"], None, ch.weight, defer=True)
                else:
                    dh_src = None
                bias_grad_acc(dg_hh, ch.bias)
                dc_next = None if first else dc_prev
            tag_amax(dg_ih_all, slot_ih)
            dx_slabs, n_x = conv_dgrad_slabs(dg_ih_all, ci.weight, g)  # the input half's data gradient: once over the window
            conv_wgrad_split_acc(dg_ih_all, rec["x_all"], None, ci.weight, defer=True)
            bias_grad_acc(dg_ih_all, ci.bias)
            return dx_slabs, n_x

        def chain_ws(L, extra):
            return [c_.ih_gates[0].weight for c_ in cells[L]] + [c_.hh_gates[0].weight for c_ in cells[L]] + extra

        # ---- frame predictor
        hp_s = None if d_hpred is None else d_hpred.view((T, B) + tuple(d_hpred.shape[1:]))
        top, n_top = layer_bwd("fp", 1, lambda t: [_src(hp_s[t], 1, g)] if hp_s is not None else [])
        low, n_low = layer_bwd("fp", 0, lambda t: [window(top, n_top, g, t)])
        dy_f_all = torch.empty((T * B, H, W, g), device=dev, dtype=torch.float32)
        slot_f = amax_slot(dev)
        tag_amax(grad_sum([_src(low, n_low, g, 0)], dy_f_all, g, slot_f), slot_f)
        dcat, n_c = conv_dgrad_slabs(dy_f_all, fw_pad, cpad)
        conv_wgrad_split_acc(dy_f_all, thin["cat"], None, fconv.weight, defer=True)  # un-pads into weight.grad
        bias_grad_acc(dy_f_all, fconv.bias)
        d_h_all = grad_sum([_src(dcat, n_c, cpad, nv)], torch.empty((T * B, H, W, g), device=dev, dtype=torch.float32), g)
        flush_deferred_wgrads_early(chain_ws("fp", [fconv.weight]))
        # ---- posterior: reparameterisation + KL gradients -> merged head -> the two layers
        dy_h = torch.empty((T * B, H, W, 2 * z), device=dev, dtype=torch.float32)
        slot = amax_slot(dev)
        call("rac_reparam_head_bwd", _pack_srcs([_src(dcat, n_c, cpad, nv + g)]), 1, ptr(ctx.lv_all), ptr(thin["eps"]),
             ptr(d_mu), ptr(d_lv), ptr(dy_h), T * M, z, ptr(slot), sp)
        tag_amax(dy_h, slot)
        dhead, n_h = conv_dgrad_slabs(dy_h, head_w, g)
        conv_wgrad_split_acc(dy_h, tag_amax(thin["h_post"], amax_one(dev)), None, head_w, defer=True)
        bias_grad_acc(dy_h, head_b)
        top, n_top = layer_bwd("post", 1, lambda t: [window(dhead, n_h, g, t)])
        low, n_low = layer_bwd("post", 0, lambda t: [window(top, n_top, g, t)])
        slot_post = amax_slot(dev)
        d_post_all = tag_amax(grad_sum([_src(low, n_low, g, 0)], torch.empty((T * B, H, W, g), device=dev, dtype=torch.float32),
                                       g, slot_post), slot_post)
        flush_deferred_wgrads_early(chain_ws("post", [head_w]))
        # ---- prior (its z is not used on this path: only its hidden state feeds the batched mu_p / logvar_p heads)
        hq_s = None if d_hprior is None else d_hprior.view((T, B) + tuple(d_hprior.shape[1:]))
        top, n_top = layer_bwd("prior", 1, lambda t: [_src(hq_s[t], 1, g)] if hq_s is not None else [])
        low, n_low = layer_bwd("prior", 0, lambda t: [window(top, n_top, g, t)])
        slot_prior = amax_slot(dev)
        d_prior_all = tag_amax(grad_sum([_src(low, n_low, g, 0)], torch.empty((T * B, H, W, g), device=dev, dtype=torch.float32),
                                        g, slot_prior), slot_prior)
        flush_deferred_wgrads_early(None)
        ctx.tape = ctx.plan = ctx.thin = None
        return (None, d_h_all, d_prior_all, d_post_all) + (None,) * (len(ctx.needs_input_grad) - 4)


class GroupNorm(torch.autograd.Function):
    """nn.GroupNorm(G, C) on an NHWC map (lstm.py:165-172), eps 1e-5."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G):
        B, H, W, Cc = x.shape
        y = torch.empty_like(x)
        stat = torch.empty((2, B, G), device=x.device, dtype=torch.float32)
        call("rac_groupnorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(stat[0]), ptr(stat[1]), B, H * W, Cc, G,
             1e-5, stream_ptr())
        ctx.save_for_backward(x, gamma, beta, stat)
        ctx.G = G
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stat = ctx.saved_tensors
        B, H, W, Cc = x.shape
        dx = torch.empty_like(x)
        want = gamma.requires_grad
        call("rac_groupnorm_bwd", ptr(dy.contiguous()), ptr(x), ptr(gamma), ptr(stat[0]), ptr(stat[1]), ptr(dx),
             ptr(grad_buffer(gamma)) if want else None, ptr(grad_buffer(beta)) if want else None, B, H * W, Cc,
             ctx.G, stream_ptr())
        return dx, None, None, None


class NormCellCore(torch.autograd.Function):
    """Gate activations and the un-normalised cell of NormConvLSTMCell (lstm.py:178-193):
    (g_ih + g_hh) -> i,f,o = sigmoid, g = tanh; c_raw = f*c_prev + i*g.  Returns (c_raw, act [.,4g])."""

    @staticmethod
    def forward(ctx, g_ih, g_hh, c_prev):
        B, H, W, g4 = g_ih.shape
        g = g4 // 4
        M = B * H * W
        two = torch.stack([g_ih, g_hh]) if (g_hh.data_ptr() - g_ih.data_ptr()) % 4 else None
        base = two if two is not None else g_ih
        stride = M * g4 if two is not None else (g_hh.data_ptr() - g_ih.data_ptr()) // 4
        zero_bias = torch.zeros(g4, device=g_ih.device)
        c_raw = torch.empty_like(c_prev)
        h_unused = torch.empty_like(c_prev)
        act = torch.empty((B, H, W, g4), device=g_ih.device, dtype=torch.float32)
        call("rac_lstm_cell_fwd", ptr(base), 2, stride, ptr(zero_bias), ptr(c_prev), ptr(h_unused), ptr(c_raw),
             ptr(act), M, g, stream_ptr())
        ctx.save_for_backward(act, c_prev)
        return c_raw, act

    @staticmethod
    def backward(ctx, dc_raw, d_act):
        act, c_prev = ctx.saved_tensors
        B, H, W, g4 = act.shape
        g = g4 // 4
        dgates = torch.empty_like(act)
        dc_prev = torch.empty_like(c_prev)
        call("rac_lstm_core_bwd", ptr(dc_raw.contiguous()) if dc_raw is not None else None,
             ptr(d_act.contiguous()) if d_act is not None else None, ptr(act), ptr(c_prev), ptr(dgates), ptr(dc_prev),
             B * H * W, g, stream_ptr())
        return dgates, dgates, dc_prev


class LstmOut(torch.autograd.Function):
    """hidden = out_gate * tanh(cell)  (lstm.py:196) with the gate read from the activation tensor."""

    @staticmethod
    def forward(ctx, act, c):
        B, H, W, g = c.shape
        h = torch.empty_like(c)
        call("rac_lstm_out_fwd", ptr(act), ptr(c), ptr(h), B * H * W, g, stream_ptr())
        ctx.save_for_backward(act, c)
        return h

    @staticmethod
    def backward(ctx, dh):
        act, c = ctx.saved_tensors
        B, H, W, g = c.shape
        d_act = torch.empty_like(act)
        dc = torch.empty_like(c)
        call("rac_lstm_out_bwd", ptr(dh.contiguous()), ptr(act), ptr(c), ptr(d_act), ptr(dc), B * H * W, g, stream_ptr())
        return d_act, dc


class Reparam(torch.autograd.Function):
    """z = eps * exp(0.5*logvar) + mu  (lstm.py:276-279)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        z = torch.empty_like(mu)
        call("rac_reparam_fwd", ptr(mu), ptr(logvar), ptr(eps), ptr(z), mu.numel(), stream_ptr())
        ctx.save_for_backward(logvar, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dz = dz.contiguous()
        dlv = torch.empty_like(dz)
        call("rac_reparam_bwd", ptr(dz), ptr(logvar), ptr(eps), ptr(dlv), dz.numel(), stream_ptr())
        return dz, dlv, None


class PackInput(torch.autograd.Function):
    """planes -> map: cat([img * (1 - zero_mask), mask, 0-pad to a multiple of 4], C) as NHWC
    (dynamics.py:578-582 + utils/image.py:5-19)."""

    @staticmethod
    def forward(ctx, img, zero_mask, mask, pad_to=0):
        B, _, H, W = img.shape
        Cm = mask.shape[1] if mask is not None else 0
        pad = pad_to - (3 + Cm) if pad_to >= 3 + Cm else pad4(3 + Cm)
        out = torch.empty((B, H, W, 3 + Cm + pad), device=img.device, dtype=torch.float32)
        call("rac_pack_input", ptr(img), ptr(zero_mask), ptr(mask), Cm, pad, ptr(out), B, H * W, stream_ptr())
        ctx.save_for_backward(zero_mask)
        ctx.C = 3 + Cm + pad
        return out

    @staticmethod
    def backward(ctx, dout):
        (zero_mask,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None, None
        dout = dout.contiguous()
        B, H, W, _ = dout.shape
        dimg = torch.empty((B, 3, H, W), device=dout.device, dtype=torch.float32)
        call("rac_unpack_grad", ptr(dout), ctx.C, ptr(zero_mask), ptr(dimg), B, H * W, stream_ptr())
        return dimg, None, None, None


def vgg_up_frozen_ok(x_low, skip, weight) -> bool:
    """The frozen decoder's conv over [upsample2(x_low) | skip] can read x_low directly (a0_up of the rows kernel)."""
    B, h, w, c0 = x_low.shape
    return (SPLIT_GEMM and weight.shape[0] >= SPLIT_MIN_COUT and 4 * h * w > 128 and skip is not None
            and tuple(skip.shape[:3]) == (B, 2 * h, 2 * w) and weight.shape[1] == c0 + skip.shape[3]
            and split_supported(2 * h, 2 * w, 3, weight.shape[1], weight.shape[0], c0))


def vgg_up_frozen(x_low, skip, weight, scale, shift) -> torch.Tensor:
    """LeakyReLU(BatchNorm_eval(conv3x3([UpsamplingNearest2d(2)(x_low) | skip]))) without the upsampled tensor."""
    return conv_forward_split(x_low, skip, weight, None, act=ACT_LEAKY, scale=scale, shift=shift, x0_up=True,
                              per_image=True)


def first_layer_ok(img, mask, weight) -> bool:
    """rac_first_layer_fwd takes this first encoder layer (frozen model only: no tape, BatchNorm folded)."""
    Cm = mask.shape[1] if mask is not None else 0
    return (img.is_cuda and weight.shape[0] == 64 and weight.shape[1] == 3 + Cm and Cm <= 5 and weight.shape[2] == 3
            and img.shape[-2] % 16 == 0 and img.shape[-1] % 16 == 0)


def first_layer_frozen(img, zero_mask, mask, weight, scale, shift) -> torch.Tensor:
    """LeakyReLU(BatchNorm_eval(conv3x3([img * (1 - zero_mask) | mask]))) as an NHWC map, straight from the planes."""
    B, _, H, W = img.shape
    Cm = mask.shape[1] if mask is not None else 0
    out = torch.empty((B, H, W, 64), device=img.device, dtype=torch.float32)
    slot = amax_slot(img.device, B)  # one maximum per image: the frozen model's scales are per image
    # the matrix pipe (per-pixel / per-channel operand scales), or exact-fp32 FMAs (RAC_FIRST_MFMA=0)
    call("rac_first_layer_fwd_split" if FIRST_MFMA else "rac_first_layer_fwd", ptr(img), ptr(zero_mask), ptr(mask), Cm,
         ptr(weight_mem(weight.detach())), ptr(scale),
         ptr(shift), ACT_LEAKY, ptr(out), ptr(slot), 1, B, H, W, 64, stream_ptr())
    return tag_amax(out, slot)


class ZeroRegion(torch.autograd.Function):
    """zero_robot_region (src/utils/image.py:5-19), out of place."""

    @staticmethod
    def forward(ctx, img, mask):
        B, _, H, W = img.shape
        out = torch.empty_like(img)
        call("rac_zero_region", ptr(img), ptr(mask), ptr(out), B, H * W, stream_ptr())
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, dout):
        (mask,) = ctx.saved_tensors
        dout = dout.contiguous()
        B, _, H, W = dout.shape
        d = torch.empty_like(dout)
        call("rac_zero_region", ptr(dout), ptr(mask), ptr(d), B, H * W, stream_ptr())
        return d, None


class Composite(torch.autograd.Function):
    """x_hat = (1 - m) * prev + m * rgb with [rgb | m] = x4 (trainer.py:406-407); x4 map -> planes."""

    @staticmethod
    def forward(ctx, x4, prev):
        B, H, W, _ = x4.shape
        out = torch.empty((B, 3, H, W), device=x4.device, dtype=torch.float32)
        call("rac_composite_fwd", ptr(x4), ptr(prev), ptr(out), B, H * W, stream_ptr())
        ctx.save_for_backward(x4, prev)
        return out

    @staticmethod
    def backward(ctx, dout):
        x4, prev = ctx.saved_tensors
        dout = dout.contiguous()
        B, H, W, _ = x4.shape
        dx4 = torch.empty_like(x4)
        dprev = torch.empty_like(prev) if ctx.needs_input_grad[1] else None
        call("rac_composite_bwd", ptr(dout), ptr(x4), ptr(prev), ptr(dx4), ptr(dprev), B, H * W, stream_ptr())
        return dx4, dprev


class ReconLoss(torch.autograd.Function):
    """mse / l1 / dontcare_mse / dontcare_l1 (losses.py:11-50) and, in the same pass, the logging
    metrics robot_mse / world_mse (losses.py:52-78).  Returns a (3,) tensor [loss, robot_mse, world_mse]."""

    @staticmethod
    def forward(ctx, pred, target, mask, batch_weight, kind, robot_weight):
        B, _, H, W = pred.shape
        per = torch.empty((B, 8), device=pred.device, dtype=torch.float32)
        out = torch.empty((3,), device=pred.device, dtype=torch.float32)
        call("rac_recon_loss_fwd", kind, ptr(pred), ptr(target), ptr(mask), float(robot_weight), ptr(batch_weight),
             ptr(per), ptr(out), B, H * W, stream_ptr())
        ctx.save_for_backward(pred, target, mask, batch_weight, per)
        ctx.kind, ctx.rw = kind, float(robot_weight)
        return out

    @staticmethod
    def backward(ctx, gout):
        pred, target, mask, bw, per = ctx.saved_tensors
        B, _, H, W = pred.shape
        g = gout.contiguous()  # only element 0 (the loss) carries gradient
        dpred = torch.empty_like(pred)
        call("rac_recon_loss_bwd", ctx.kind, ptr(pred), ptr(target), ptr(mask), ctx.rw, ptr(bw), ptr(per), ptr(g),
             ptr(dpred), B, H * W, stream_ptr())
        return dpred, None, None, None, None, None


class KLLoss(torch.autograd.Function):
    """kl_criterion (losses.py:97-106)."""

    @staticmethod
    def forward(ctx, mu1, lv1, mu2, lv2, bs):
        out = torch.empty((1,), device=mu1.device, dtype=torch.float32)
        part = torch.empty((1,), device=mu1.device, dtype=torch.float64)
        call("rac_kl_fwd", ptr(mu1), ptr(lv1), ptr(mu2), ptr(lv2), mu1.numel(), bs, ptr(part), ptr(out), stream_ptr())
        ctx.save_for_backward(mu1, lv1, mu2, lv2)
        ctx.bs = bs
        return out

    @staticmethod
    def backward(ctx, gout):
        mu1, lv1, mu2, lv2 = ctx.saved_tensors
        outs = [torch.empty_like(mu1) for _ in range(4)]
        call("rac_kl_bwd", ptr(mu1), ptr(lv1), ptr(mu2), ptr(lv2), ptr(gout.contiguous()), mu1.numel(), ctx.bs,
             *[ptr(o) for o in outs], stream_ptr())
        return outs[0], outs[1], outs[2], outs[3], None


def to_map(planes: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) logical tensor -> contiguous (B,H,W,C) map (zero-copy for channels_last inputs)."""
    return planes.permute(0, 2, 3, 1).contiguous()


def to_planes_view(m: torch.Tensor) -> torch.Tensor:
    """(B,H,W,C) map -> logical (B,C,H,W) view (channels_last strides, no copy)."""
    return m.permute(0, 3, 1, 2)
