"""HIP keypoint-bottleneck kernels vs the reference's golden vectors and the CPU oracle.

Tolerance (op level): atol 1e-6 + rtol 1e-5 in fp32.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-6, 1e-5
CASES = ['c4', 'c5', 'c1', 'pong', 'ragged', 'centre5', 'corner5', 'spike16', 'uniform', 'pm80', 'one_pixel']


def close(a, b, atol=ATOL, rtol=RTOL):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b)
    assert np.all(err <= atol + rtol * np.abs(b)), f'max err {err.max():.3e}'


@pytest.mark.parametrize('tag', CASES)
def test_spatial_logsoftmax_golden(golden, tag):
    from keypoints_amd.models import functional as MF
    z = golden('bottleneck.npz')
    h = torch.tensor(z[f'{tag}.h'], device='cuda', requires_grad=True)
    k, (p, q) = MF.spacial_logsoftmax(h, probs=True)
    close(k, z[f'{tag}.k']); close(p, z[f'{tag}.p']); close(q, z[f'{tag}.q'])
    (k * torch.tensor(z[f'{tag}.wk'], device='cuda')).sum().backward()
    close(h.grad, z[f'{tag}.dh'], atol=2e-6)
    k2 = MF.spacial_softmax(h.detach())
    close(k2, z[f'{tag}.k_soft'])
    assert MF.spacial_logsoftmax(h.detach()).shape == (h.shape[0], h.shape[1], 2)   # tests/tests.py:91-93


def test_spike_keypoint_value(golden):
    """tests/tests.py:237-241: a 20.0 spike at [0,15] of a 16x16 map localises only to (0.4327, 0.5673)."""
    from keypoints_amd.models import functional as MF
    h = torch.zeros(1, 1, 16, 16, device='cuda'); h[0, 0, 0, 15] = 20.0
    k = MF.spacial_logsoftmax(h)[0, 0].cpu()
    assert abs(k[0].item() - 0.4327) < 1e-4 and abs(k[1].item() - 0.5673) < 1e-4


@pytest.mark.parametrize('tag', CASES + ['nodes'])
def test_gaussian_like_golden(golden, tag):
    from keypoints_amd.models import functional as MF
    z = golden('bottleneck.npz')
    k = torch.tensor(z[f'{tag}.k'], device='cuda', requires_grad=True)
    H, W = z[f'{tag}.m'].shape[2:]
    m = MF.gaussian_like_function(k, H, W)
    close(m, z[f'{tag}.m'])
    (m * torch.tensor(z[f'{tag}.wm'], device='cuda')).sum().backward()
    close(k.grad, z[f'{tag}.dk'], atol=1e-4, rtol=1e-4)   # sums of H*W terms with 1/d up to 1e3
    if tag != 'nodes':
        close(MF.point_map(k.detach(), H, W), z[f'{tag}.point_map'])
    else:
        close(MF.gaussian_like_function(k.detach(), 9, 12, sigma=0.25, eps=1e-4), z['nodes.m_s025'])


def test_transport_max_golden(golden):
    from keypoints_amd.models import functional as MF
    z = golden('transport.npz')
    a = [torch.tensor(z[n], device='cuda', requires_grad=True) for n in ('phi_s', 'phi_t', 'm_s', 'm_t')]
    phi, ms, mt = MF.transport_max(*a)
    close(phi, z['max.phi']); close(ms, z['max.mask_s']); close(mt, z['max.mask_t'])
    (phi * torch.tensor(z['w'], device='cuda')).sum().backward()
    for t, n in zip(a, ('d_phi_s', 'd_phi_t', 'd_m_s', 'd_m_t')):
        close(t.grad, z['max.' + n], atol=1e-5)


@pytest.mark.parametrize('mode', ['max', 'sum_and_clamp', 'loop', 'pretrained_network'])
def test_combine_modes_in_net(golden, mode):
    """All four combine modes of TransporterNet.forward (transporter.py:41-60) through our module."""
    from keypoints_amd.models import transporter
    z = golden('transport.npz')
    net = transporter.TransporterNet.__new__(transporter.TransporterNet)
    torch.nn.Module.__init__(net)
    net.combine_method = mode
    net.decoder = lambda x: x
    t = {n: torch.tensor(z[n], device='cuda') for n in ('phi_s', 'phi_t', 'm_s', 'm_t')}
    ms, mt = (t['m_s'], t['m_t']) if mode != 'pretrained_network' else (t['m_s'][:, :1], t['m_t'][:, :1])
    k = torch.zeros(2, 5, 2, device='cuda')
    # stand-ins for the sub-networks: feature / keypoint return the stored tensors, key2map the stored maps
    feats, maps = iter([t['phi_s'], t['phi_t']]), iter([ms, mt])
    net.feature = lambda x: next(feats)
    net.keypoint = lambda x: torch.zeros(2, 5, 8, 8, device='cuda')
    net.ssm = lambda h, probs=False: (k, None)
    net.key2map = lambda kk, height, width: next(maps)
    res = net.forward(None, None)
    close(res[1], z[f'{mode}.phi']); close(res[6], z[f'{mode}.mask_s']); close(res[7], z[f'{mode}.mask_t'])


def test_full_size_properties():
    """BASELINE-size (B=128, K=10, 16x16 and K=30, 32x32) invariants: keypoints in [0,1], marginals sum
    to 1, a uniform map gives (0.5, 0.5), the rendered peak is exp(-sqrt(eps)/(2 sigma^2)) at a node."""
    from keypoints_amd.models import functional as MF
    torch.manual_seed(7)
    for (B, K, H) in ((128, 10, 16), (64, 30, 32)):
        h = torch.randn(B, K, H, H, device='cuda') * 4
        k, (p, q) = MF.spacial_logsoftmax(h, probs=True)
        assert k.min() >= 0 and k.max() <= 1
        assert (p.sum(2) - 1).abs().max() < 1e-5 and (q.sum(2) - 1).abs().max() < 1e-5
        ku = MF.spacial_logsoftmax(torch.zeros(2, 3, H, H, device='cuda'))
        assert (ku - 0.5).abs().max() < 1e-6
        m = MF.gaussian_like_function(k, H, H)
        assert m.max() <= np.exp(-np.sqrt(1e-6) * 50) + 1e-6 and m.min() >= 0
        # oracle on a slice
        from oracle import kp_oracle as O
        ok, _ = O.spatial_logsoftmax(h[:4].cpu())
        close(k[:4], ok.numpy())
        # the map is rendered from this implementation's keypoints: a 1-ulp (2.4e-7) difference of a coordinate
        # against the oracle's moves exp(-50 d) by up to 50*sqrt(2)*ulp = 1.7e-5 relative
        close(m[:4], O.gaussian_like(ok, H, H).numpy(), atol=2e-6, rtol=1e-4)


class _DropGrad(torch.autograd.Function):
    """A downstream op that hands back an UNDEFINED gradient (None) for its input."""

    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return None


@pytest.mark.parametrize('mode', ['softmax', 'max', 'sum_and_clamp', 'loop'])
def test_undefined_incoming_gradient_is_zero_not_an_error(mode):
    """The ops run with set_materialize_grads(False): an undefined (None) gradient of their one differentiable output must come back
    as an undefined gradient of the inputs -- torch's meaning of "zero" -- not as an AttributeError on None."""
    from keypoints_amd.models import functional as MF
    g = torch.Generator().manual_seed(0)
    if mode == 'softmax':
        h = torch.randn(2, 3, 8, 8, generator=g).cuda().requires_grad_(True)
        k = MF.spacial_logsoftmax(h)
        (_DropGrad.apply(k).sum() + 0.0 * h.sum()).backward()
        assert float(h.grad.abs().max()) == 0.0
        return
    phi_s, phi_t = (torch.randn(2, 4, 8, 8, generator=g).cuda().requires_grad_(True) for _ in range(2))
    m_s, m_t = (torch.rand(2, 3, 8, 8, generator=g).cuda().requires_grad_(True) for _ in range(2))
    phi, _, _ = MF.transport(phi_s, phi_t, m_s, m_t, mode)
    (_DropGrad.apply(phi).sum() + 0.0 * (phi_s.sum() + phi_t.sum() + m_s.sum() + m_t.sum())).backward()
    for t in (phi_s, phi_t, m_s, m_t):
        assert float(t.grad.abs().max()) == 0.0
