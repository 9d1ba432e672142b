"""Thin-plate-spline and rotation warps with the reference's names (keypoints/tps.py), backed by the HIP
resampling kernels of kp_warp.hip: the sampling position is computed per output pixel on the device, so no
[N][H][W][2] grid is built on the host and shipped over PCIe as the reference does (tps.py:81-84,130,159-165).

Random parameters are drawn on the CPU with the same torch RNG calls in the same order as the reference
(tps.py:122-125), so a seed reproduces the reference's augmentation.
"""
import torch

from ._lib import call, ptr
from .runtime import f32c, require_device, stream_ptr


def tps_sample_params(batch_size, num_control_points, var=0.05):
    theta = torch.randn(batch_size, num_control_points + 3, 2) * var
    cnt_points = torch.rand(batch_size, num_control_points, 2)
    return theta, cnt_points


def tps_transform(x, theta, cnt_points):
    """F.grid_sample(x, tps_grid(theta, cnt_points, x.shape), padding_mode='zeros')   (tps.py:128-131)"""
    require_device(x)
    x = f32c(x)
    N, C, H, W = x.shape
    T = cnt_points.shape[-2]
    params = theta.shape[1]
    if params not in (T + 3, T + 2):
        raise ValueError(f'theta has {params} rows for {T} control points (expected T+3 or T+2)')
    th = f32c(theta.to(x.device))
    ct = f32c(cnt_points.to(x.device))
    out = torch.empty_like(x)
    call('kp_warp_tps', ptr(x), ptr(th), ptr(ct), ptr(out), N, C, H, W, T, int(params == T + 2), int(ct.dim() == 3),
         stream_ptr())
    return out


def rotate_affine_grid_multi(x, theta):
    """Per-sample rotation by `theta` radians through affine_grid + grid_sample semantics (tps.py:154-166).  The rotation
    matrices are formed on the device from the angles (kp_warp_rotate): nothing is built on the host, and angles that are
    already on the device (data_augments.TpsAndRotate ships a whole draw in one copy) cost no transfer at all."""
    require_device(x)
    x = f32c(x)
    N, C, H, W = x.shape
    ang = f32c(theta.detach().reshape(-1).to(x.device))
    if ang.numel() != N:
        raise ValueError(f'{ang.numel()} angles for a batch of {N}')
    out = torch.empty_like(x)
    call('kp_warp_rotate', ptr(x), ptr(ang), ptr(out), N, C, H, W, stream_ptr())
    return out


class RandomTPSTransform(object):
    def __init__(self, num_control=4, variance=0.05):
        self.num_control = num_control
        self.var = variance

    def __call__(self, x):
        theta, cnt_points = tps_sample_params(x.size(0), self.num_control, self.var)
        return tps_transform(x, theta, cnt_points)


class RandRotate(object):
    def __init__(self, max=0.2):
        self.max = max

    def __call__(self, x):
        theta = (torch.rand(x.size(0)) * 2 - 1) * self.max
        return rotate_affine_grid_multi(x, theta)
