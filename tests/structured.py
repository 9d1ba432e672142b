"""Structured synthetic image pairs for the conditioned-network parity test (test infrastructure; no reference code).

A pair shares a smooth low-frequency background (a sum of a few random 2-D cosines per colour channel) and differs in the
positions of a handful of soft-edged coloured blobs -- the structure the Transporter is built for (static scene, moving
objects: SURVEY 1), in [0, 1] like the reference's `ToTensor` CelebA frames (keypoints/ds/datasets.py:297-300).
The random parameters come from the torch generator passed in, on the CPU, so a seed fixes the scene on every box."""
import math

import torch


def structured_pairs(B, size, gen, blobs=4, channels=3, device='cpu'):
    """-> (xs, xt), each [B, channels, size, size] fp32 on `device`.  The random parameters (a few hundred numbers) are drawn on
    the CPU from `gen`; the images are rendered on `device`."""
    ax = torch.linspace(0, 1, size, device=device)
    yy, xx = torch.meshgrid(ax, ax, indexing='ij')
    u = lambda *s: torch.rand(*s, generator=gen).to(device)
    # background: 4 cosines per channel, amplitudes falling with frequency
    fx, fy = u(B, channels, 4) * 3, u(B, channels, 4) * 3
    ph, am = u(B, channels, 4) * 2 * math.pi, u(B, channels, 4) * 0.25
    bg = 0.5 + (am[..., None, None] * torch.cos(2 * math.pi * (fx[..., None, None] * xx + fy[..., None, None] * yy)
                                                  + ph[..., None, None])).sum(2) * 0.5
    col = u(B, blobs, channels)
    rad = 0.04 + 0.06 * u(B, blobs)

    def render(pos):
        img = bg.clone()
        for j in range(blobs):
            d2 = (yy[None] - pos[:, j, 0, None, None]) ** 2 + (xx[None] - pos[:, j, 1, None, None]) ** 2
            a = torch.sigmoid((rad[:, j, None, None] ** 2 - d2) / (0.02 * rad[:, j, None, None]))   # soft edge
            img = img * (1 - a[:, None]) + col[:, j, :, None, None] * a[:, None]
        return img.clamp_(0, 1)
    ps = 0.1 + 0.8 * u(B, blobs, 2)
    pt = (ps + 0.25 * (u(B, blobs, 2) - 0.5)).clamp_(0.05, 0.95)
    return render(ps).contiguous(), render(pt).contiguous()
