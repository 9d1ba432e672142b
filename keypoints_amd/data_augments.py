"""Input augmentation with the reference's surface (keypoints/data_augments.py:6-38): `nop` and `TpsAndRotate`
turn a batch into the (x, x_, loss_mask) triple the train loop feeds to the model (keypoints.py:72-79,
configs/keypoints_celeba.yaml:13-16).  Every warp is one HIP resampling pass (keypoints_amd.tps); the image and its
all-ones validity mask go through the same two (TPS, rotation) stages, so the mask marks what survived both.
"""
import torch

from . import tps as _tps


def rand_peturb_params(batch_items, tps_cntl_pts, tps_variance, max_rotate):
    """One random (TPS weights, control points, rotation angle) draw per sample; same torch RNG calls, in the same
    order, as data_augments.py:6-10, so a seed reproduces the reference's augmentation."""
    weights, points = _tps.tps_sample_params(batch_items, tps_cntl_pts, tps_variance)
    angles = max_rotate * (2 * torch.rand(batch_items) - 1)
    return weights, points, angles


def peturb(x, tps_theta, cntl_pts, theta_rotate):
    """TPS warp followed by a rotation (data_augments.py:13-16)."""
    warped = _tps.tps_transform(x, tps_theta, cntl_pts)
    return _tps.rotate_affine_grid_multi(warped, theta_rotate)


def _ship_once(params, device):
    """All tensors of one random draw (TPS weights, control points, angles: a few KB) to `device` in ONE non-blocking copy
    through a pinned staging buffer, returned as views of the device buffer.  The reference -- and round 2 here -- uploaded every
    tensor for every warp (image and mask: 6 pageable, i.e. host-blocking, copies per draw)."""
    if device.type != 'cuda':
        return params
    flat = torch.cat([p.detach().reshape(-1).float() for p in params])
    stage = torch.empty(flat.numel(), dtype=torch.float32, pin_memory=True)
    stage.copy_(flat)
    dev = stage.to(device, non_blocking=True)
    out, off = [], 0
    for p in params:
        out.append(dev[off:off + p.numel()].view(p.shape))
        off += p.numel()
    return tuple(out)


def nop(*data):
    """No augmentation: (x, x_) pass through, no loss mask."""
    return data[0], data[1], None


class TpsAndRotate(object):
    """x -> (peturb_1(x), peturb_2(peturb_1(x)), mask) with two independent random draws."""

    def __init__(self, data_aug_tps_cntl_pts, data_aug_tps_variance, data_aug_max_rotate):
        self.tps_cntl_pts = data_aug_tps_cntl_pts
        self.tps_variance = data_aug_tps_variance
        self.max_rotate = data_aug_max_rotate

    def _draw(self, n):
        return rand_peturb_params(n, self.tps_cntl_pts, self.tps_variance, self.max_rotate)

    def __call__(self, *data):
        image = data[0]
        valid = torch.ones_like(image)
        stages = []
        for _ in range(2):                      # draw, warp the image, warp the mask -- in the reference's order
            params = _ship_once(self._draw(image.size(0)), image.device)
            image = peturb(image, *params)
            valid = peturb(valid, *params)
            stages.append(image)
        return stages[0], stages[1], valid
