"""`nop` / `TpsAndRotate` with the reference's surface (keypoints/data_augments.py:6-38): the augmentation that
turns a batch x into (x, x_, loss_mask) for the KeyNet config (configs/keypoints_celeba.yaml:13-16)."""
import torch

from .tps import rotate_affine_grid_multi, tps_sample_params, tps_transform


def rand_peturb_params(batch_items, tps_cntl_pts, tps_variance, max_rotate):
    theta_tps, cntl_pts = tps_sample_params(batch_items, tps_cntl_pts, tps_variance)
    theta_rotate = (torch.rand(batch_items) * 2 - 1) * max_rotate
    return theta_tps, cntl_pts, theta_rotate


def peturb(x, tps_theta, cntl_pts, theta_rotate):
    return rotate_affine_grid_multi(tps_transform(x, tps_theta, cntl_pts), theta_rotate)


def nop(*data):
    return data[0], data[1], None


class TpsAndRotate(object):
    def __init__(self, data_aug_tps_cntl_pts, data_aug_tps_variance, data_aug_max_rotate):
        self.tps_cntl_pts, self.tps_variance, self.max_rotate = \
            data_aug_tps_cntl_pts, data_aug_tps_variance, data_aug_max_rotate

    def __call__(self, *data):
        x = data[0]
        loss_mask = torch.ones(x.shape, dtype=x.dtype, device=x.device)
        p1 = rand_peturb_params(x.size(0), self.tps_cntl_pts, self.tps_variance, self.max_rotate)
        x = peturb(x, *p1)
        loss_mask = peturb(loss_mask, *p1)
        p2 = rand_peturb_params(x.size(0), self.tps_cntl_pts, self.tps_variance, self.max_rotate)
        x_ = peturb(x, *p2)
        loss_mask = peturb(loss_mask, *p2)
        return x, x_, loss_mask
