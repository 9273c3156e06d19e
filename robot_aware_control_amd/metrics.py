"""PSNR / SSIM of the evaluation path (reference src/utils/metrics.py:45-78), one fused HIP launch."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def _run(img1, img2, mask=None, want_map=False):
    if not img1.is_cuda:
        raise _lib.RacError("metrics run on the GPU only (no CPU fallback)")
    a, b = img1.to(torch.float32).contiguous(), img2.to(torch.float32).contiguous()
    N, C, H, W = a.shape
    assert C == 3 and b.shape == a.shape
    acc = torch.zeros((2, N), device=a.device, dtype=torch.float32)
    ssim_map = torch.empty_like(a) if want_map else None
    m = None if mask is None else mask.to(torch.float32).contiguous()
    _lib.call("rac_psnr_ssim", a.data_ptr(), b.data_ptr(), _lib.ptr(m), acc[0].data_ptr(), acc[1].data_ptr(),
              _lib.ptr(ssim_map), N, H, W, _lib.stream_ptr())
    return acc, ssim_map, 3 * H * W


@torch.no_grad()
def psnr(estimates, targets, data_dims=3):
    """Per-sample PSNR of two [0,1] batches; like the reference both are first mapped to (x+1)/2
    (src/utils/metrics.py:61-78).  Inputs are clamped to [0,1] as the eval loop does (trainer.py:685)."""
    acc, _, n = _run(estimates, targets)
    return 10.0 * torch.log(1.0 / (acc[0] / n)) / np.log(10)


@torch.no_grad()
def ssim(img1, img2, window_size=11):
    """SSIM map as a numpy array (N,3,H,W) (src/utils/metrics.py:45-58)."""
    assert window_size == 11
    return _run(img1, img2, want_map=True)[1].cpu().numpy()


@torch.no_grad()
def masked_psnr_ssim(pred, target, mask):
    """(per-sample PSNR tensor, mean SSIM tensor) of zero_robot_region(mask, .) versions of both frames --
    the fused form of trainer.py:681-693; no host sync."""
    acc, _, n = _run(target, pred, mask)
    return 10.0 * torch.log(1.0 / (acc[0] / n)) / np.log(10), acc[1].sum() / (acc.shape[1] * n)
