"""zero_robot_region (reference src/utils/image.py:5-19): GPU tensors go through the HIP kernel,
numpy arrays are handled on the host exactly like the reference's numpy branch."""
import numpy as np
import torch

from . import ops


def zero_robot_region(mask, image, inplace=False):
    """Set the robot region of `image` to zero.  Tensor: image (B,3,H,W), mask (B,1,H,W) {0,1}."""
    if isinstance(mask, torch.Tensor):
        if not image.is_cuda:
            raise ops._lib.RacError("zero_robot_region: tensor inputs must live on the GPU (no CPU fallback)")
        out = ops.ZeroRegion.apply(image.contiguous(), mask.to(torch.float32).contiguous())
        if inplace:
            image.copy_(out)
            return image
        return out
    robot_mask = mask.astype(bool)
    if not inplace:
        image = image.copy()
    image[robot_mask] = 0
    return image
