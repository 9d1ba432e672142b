"""TPS + rotation augmentation (keypoints/tps.py, keypoints/data_augments.py).
CPU: the oracle restatement against the vectors generated from the reference (tools/make_golden.py tps).
GPU: the HIP resampling kernels (kp_warp.hip) through keypoints_amd.tps / data_augments against the same vectors.
Tolerance: the reference's own fp32-vs-fp64 noise floor on these vectors is 1.7e-6 .. 2.3e-6 (an image in [0, 1]
resampled at a position that itself carries fp32 rounding; tools/probes/tpserr.py); the HIP kernels sit at the same
distance from the fp64 result, 3e-6 from the fp32 one.  Stated tolerance: atol 1e-5 (about 5x the floor)."""
import numpy as np
import pytest
import torch

TAGS = ['a', 'b']


def _params(z, tag, i):
    return (torch.tensor(z[f'{tag}.theta{i}']), torch.tensor(z[f'{tag}.ctrl{i}']), torch.tensor(z[f'{tag}.rot{i}']))


@pytest.mark.parametrize('tag', TAGS)
def test_oracle_tps_matches_reference_vectors(golden, tag):
    from oracle import kp_oracle as O
    z = golden('tps.npz')
    x = torch.tensor(z[f'{tag}.x'])
    p0, p1 = _params(z, tag, 0), _params(z, tag, 1)
    assert np.abs(O.tps_grid(p0[0], p0[1], x.shape).numpy() - z[f'{tag}.grid']).max() <= 1e-6
    assert np.abs(O.tps_transform(x, p0[0], p0[1]).numpy() - z[f'{tag}.warp']).max() <= 1e-6
    assert np.abs(O.rotate_affine_grid_multi(x, p0[2]).numpy() - z[f'{tag}.rot']).max() <= 1e-6
    x1, x2, m = O.tps_and_rotate(x, p0, p1)
    for got, key in ((x1, 'x1'), (x2, 'x2'), (m, 'mask')):
        assert np.abs(got.numpy() - z[f'{tag}.{key}']).max() <= 1e-6
    red = O.tps_transform(x, torch.tensor(z[f'{tag}.theta_red']), p0[1])
    assert np.abs(red.numpy() - z[f'{tag}.warp_red']).max() <= 1e-6


def test_parameter_draws_follow_the_reference_rng_order(golden):
    from keypoints_amd.data_augments import rand_peturb_params
    z = golden('tps.npz')
    torch.manual_seed(5)
    th, ct, ro = rand_peturb_params(4, 4, 0.05, 0.1)
    assert np.array_equal(th.numpy(), z['draw.theta']) and np.array_equal(ct.numpy(), z['draw.ctrl'])
    assert np.array_equal(ro.numpy(), z['draw.rot'])


def test_nop_augment():
    from keypoints_amd.data_augments import nop
    a, b = torch.zeros(1), torch.ones(1)
    assert nop(a, b) == (a, b, None)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', TAGS)
def test_hip_tps_and_rotate_match_reference_vectors(golden, tag):
    from keypoints_amd import data_augments as DA, tps as TPS
    z = golden('tps.npz')
    x = torch.tensor(z[f'{tag}.x']).cuda()
    p0, p1 = _params(z, tag, 0), _params(z, tag, 1)
    tol = 1e-5
    assert np.abs(TPS.tps_transform(x, p0[0], p0[1]).cpu().numpy() - z[f'{tag}.warp']).max() <= tol
    assert np.abs(TPS.rotate_affine_grid_multi(x, p0[2]).cpu().numpy() - z[f'{tag}.rot']).max() <= tol
    red = TPS.tps_transform(x, torch.tensor(z[f'{tag}.theta_red']), p0[1])
    assert np.abs(red.cpu().numpy() - z[f'{tag}.warp_red']).max() <= tol
    shared = TPS.tps_transform(x, p0[0], p0[1][0])          # one set of control points for the whole batch
    assert np.abs(shared[0].cpu().numpy() - z[f'{tag}.warp'][0]).max() <= tol
    x1 = DA.peturb(x, *p0)
    x2 = DA.peturb(x1, *p1)
    m = DA.peturb(DA.peturb(torch.ones_like(x), *p0), *p1)
    for got, key in ((x1, 'x1'), (x2, 'x2'), (m, 'mask')):
        assert np.abs(got.cpu().numpy() - z[f'{tag}.{key}']).max() <= 2 * tol   # two / four chained resamplings


@pytest.mark.gpu
def test_rotation_from_device_angles_equals_the_host_built_matrices():
    """kp_warp_rotate (matrix [[cos, sin, 0], [-sin, cos, 0]] formed in the kernel from angles on the device) against
    kp_warp_affine fed the matrices tps.py:159-165 builds on the CPU; and TpsAndRotate with its one pinned upload per draw
    against peturb() called with CPU parameters (the two differ by how the parameters travel, not by a single bit of them)."""
    from keypoints_amd import _lib, data_augments as DA
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(3)
    x = torch.rand(5, 3, 37, 50, generator=g).cuda()
    ang = (torch.rand(5, generator=g) * 2 - 1) * 0.3
    mat = torch.zeros(5, 2, 3)
    mat[:, 0, 0] = torch.cos(ang); mat[:, 0, 1] = torch.sin(ang)
    mat[:, 1, 0] = -torch.sin(ang); mat[:, 1, 1] = torch.cos(ang)
    a, b = torch.empty_like(x), torch.empty_like(x)
    call('kp_warp_affine', ptr(x), ptr(mat.cuda()), ptr(a), 5, 3, 37, 50, stream_ptr())
    call('kp_warp_rotate', ptr(x), ptr(ang.cuda()), ptr(b), 5, 3, 37, 50, stream_ptr())
    assert float((a - b).abs().max()) <= 2e-6      # cosf / sinf on the device vs torch's on the CPU: an ulp of the matrix
    torch.manual_seed(11)
    aug = DA.TpsAndRotate(4, 0.05, 0.1)
    x1, x2, mask = aug(x, x)
    torch.manual_seed(11)
    p1 = DA.rand_peturb_params(5, 4, 0.05, 0.1)
    p2 = DA.rand_peturb_params(5, 4, 0.05, 0.1)
    r1 = DA.peturb(x, *p1)
    r2 = DA.peturb(r1, *p2)
    rm = DA.peturb(DA.peturb(torch.ones_like(x), *p1), *p2)
    assert torch.equal(x1, r1) and torch.equal(x2, r2) and torch.equal(mask, rm)


@pytest.mark.gpu
def test_hip_tps_full_size_properties():
    """128x128x3 batch of 128 (config 3 shape): identity parameters reproduce the image up to the half-pixel
    resampling of align_corners=False (interior of a constant image stays constant; mask of ones stays <= 1)."""
    from keypoints_amd.data_augments import TpsAndRotate
    from keypoints_amd import tps as TPS
    x = torch.full((128, 3, 128, 128), 0.5, device='cuda')
    theta = torch.zeros(128, 7, 2)
    ctrl = torch.rand(128, 4, 2)
    y = TPS.tps_transform(x, theta, ctrl)
    assert (y[:, :, 2:-2, 2:-2] - 0.5).abs().max().item() <= 1e-6
    torch.manual_seed(0)
    xa, xb, mask = TpsAndRotate(4, 0.05, 0.1)(torch.rand(128, 3, 128, 128, device='cuda'))
    assert xa.shape == xb.shape == mask.shape == (128, 3, 128, 128)
    assert torch.isfinite(xa).all() and torch.isfinite(xb).all()
    assert mask.min().item() >= 0.0 and mask.max().item() <= 1.0 + 1e-5
    assert 0.0 <= xa.min().item() and xa.max().item() <= 1.0 + 1e-5
