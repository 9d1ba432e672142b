"""kp_keypoint_bottleneck_fwd / _bwd (one launch per direction) against the golden vectors, the piecewise kernels and
the oracle's autograd.  Shapes: 16x16/K=10 (config 4), 32x32/K=30 (config 5), 84x84/K=4 (config 2), a ragged 7x13 map
(scalar path: H*W not a multiple of 4), and the BASELINE batch sizes.

Tolerance (op level): atol 1e-6 + rtol 1e-5 in fp32 for the forward; the forward must also equal the
piecewise kernels bit for bit (same expressions, same summation order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(5, 64, 10, 16, 16), (3, 64, 30, 32, 32), (3, 16, 4, 84, 84), (2, 6, 5, 7, 13), (2, 3, 3, 32, 32), (1, 1, 1, 1, 1)]


def close(a, b, atol=1e-6, rtol=1e-5):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b)
    assert np.all(err <= atol + rtol * np.abs(b)), f'max err {err.max():.3e}'


def _inputs(B, C, K, H, W, seed=0, scale=3.0):
    g = torch.Generator().manual_seed(seed)
    heat_s = torch.randn(B, K, H, W, generator=g) * scale
    heat_t = torch.randn(B, K, H, W, generator=g) * scale
    phi_s = torch.randn(B, C, H, W, generator=g)
    phi_t = torch.randn(B, C, H, W, generator=g)
    return heat_s, heat_t, phi_s, phi_t


def _piecewise(heat_s, heat_t, phi_s, phi_t):
    from keypoints_amd.models import functional as MF
    H, W = phi_t.shape[2:]
    with torch.no_grad():
        ks = MF.spacial_logsoftmax(heat_s)
        ms = MF.gaussian_like_function(ks, H, W)
    kt, (p, q) = MF.spacial_logsoftmax(heat_t, probs=True)
    mt = MF.gaussian_like_function(kt, H, W)
    phi, mask_s, mask_t = MF.transport_max(phi_s, phi_t, ms, mt)
    return phi, kt, (p, q), mt, mask_s, mask_t


@pytest.mark.parametrize('shape', SHAPES)
def test_fused_forward_is_bitwise_the_piecewise_chain(shape):
    from keypoints_amd.models import functional as MF
    a = [t.cuda() for t in _inputs(*shape)]
    phi, k, (p, q), m, ms, mt = MF.keypoint_bottleneck(*a)
    rphi, rk, (rp, rq), rm, rms, rmt = _piecewise(*a)
    for got, ref in ((phi, rphi), (k, rk), (p, rp), (q, rq), (m, rm), (ms, rms), (mt, rmt)):
        assert got.shape == ref.shape and torch.equal(got, ref)


@pytest.mark.parametrize('tag', ['c4', 'c5', 'c1', 'pong', 'ragged', 'pm80', 'spike16', 'uniform', 'one_pixel'])
def test_fused_forward_golden(golden, tag):
    """Heat-maps of tests/golden/bottleneck.npz (captured from the reference) on the target branch."""
    from keypoints_amd.models import functional as MF
    from oracle import kp_oracle as O
    z = golden('bottleneck.npz')
    h = torch.tensor(z[f'{tag}.h'])
    B, K, H, W = h.shape
    g = torch.Generator().manual_seed(3)
    hs = torch.randn(B, K, H, W, generator=g) * 2
    phi_s, phi_t = torch.randn(B, 8, H, W, generator=g), torch.randn(B, 8, H, W, generator=g)
    phi, k, (p, q), m, ms, mt = MF.keypoint_bottleneck(hs.cuda(), h.cuda(), phi_s.cuda(), phi_t.cuda())
    close(k, z[f'{tag}.k']); close(p, z[f'{tag}.p']); close(q, z[f'{tag}.q'])
    close(m, z[f'{tag}.m'], atol=2e-6, rtol=1e-4)   # rendered from THIS k: see the comment below
    oks, _ = O.spatial_logsoftmax(hs)
    ophi, oms, omt = O.transport(phi_s, phi_t, O.gaussian_like(oks, H, W), torch.tensor(z[f'{tag}.m']), 'max')
    # maps are rendered from this implementation's keypoints: 1 ulp of a coordinate moves exp(-50 d) by 1.7e-5 relative;
    # the source keypoints come from torch's CPU sums here, which differ from the kernel's by a few ulps over 84 terms
    close(ms, oms, atol=1e-5, rtol=2e-4); close(mt, omt, atol=2e-6, rtol=1e-4)
    close(phi, ophi, atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize('shape', SHAPES[:5])
@pytest.mark.parametrize('extra', [False, True])
def test_fused_backward_vs_oracle_autograd(shape, extra):
    """d_heat_t and d_phi_t against torch autograd through the oracle (fp64 on the CPU); `extra` adds direct gradients
    on k_t and m_t (a loss that also looks at the keypoints / maps), which enter the chain inside the kernel."""
    from keypoints_amd.models import functional as MF
    from oracle import kp_oracle as O
    B, C, K, H, W = shape
    heat_s, heat_t, phi_s, phi_t = _inputs(*shape, seed=5)
    g = torch.Generator().manual_seed(6)
    w_phi = torch.randn(B, C, H, W, generator=g)
    w_k = torch.randn(B, K, 2, generator=g)
    w_m = torch.randn(B, K, H, W, generator=g)

    ht = heat_t.cuda().requires_grad_(True)
    pt = phi_t.cuda().requires_grad_(True)
    phi, k, _, m, _, _ = MF.keypoint_bottleneck(heat_s.cuda(), ht, phi_s.cuda(), pt)
    loss = (phi * w_phi.cuda()).sum()
    if extra:
        loss = loss + (k * w_k.cuda()).sum() + (m * w_m.cuda()).sum()
    loss.backward()

    oh = heat_t.double().requires_grad_(True)
    op = phi_t.double().requires_grad_(True)
    with torch.no_grad():
        oks, _ = O.spatial_logsoftmax(heat_s.double())
        oms = O.gaussian_like(oks, H, W)
    okt, _ = O.spatial_logsoftmax(oh)
    omt = O.gaussian_like(okt, H, W)
    ophi, _, _ = O.transport(phi_s.double(), op, oms, omt, 'max')
    oloss = (ophi * w_phi.double()).sum()
    if extra:
        oloss = oloss + (okt * w_k.double()).sum() + (omt * w_m.double()).sum()
    oloss.backward()
    close(pt.grad, op.grad, atol=1e-5, rtol=1e-5)
    scale = float(oh.grad.abs().max())
    assert float((ht.grad.cpu().double() - oh.grad).abs().max()) <= 2e-4 * scale + 1e-7   # sums of H*W terms with 1/d up to 1e3


def test_fused_matches_piecewise_at_baseline_sizes():
    """BASELINE batches: config 4 (128 pairs, 16x16, K=10, z=64), config 2 (256, 84x84, K=4, z=16), config 5
    (256, 32x32, K=30, z=64): forward bitwise equal to the piecewise kernels, backward equal within fp32 summation
    order, every output written (NaN prefill)."""
    from keypoints_amd.models import functional as MF
    for shape in ((128, 64, 10, 16, 16), (256, 16, 4, 84, 84), (256, 64, 30, 32, 32)):
        a = [t.cuda() for t in _inputs(*shape, seed=9)]
        a[1].requires_grad_(True); a[3].requires_grad_(True)
        phi, k, (p, q), m, ms, mt = MF.keypoint_bottleneck(*a)
        w = torch.randn_like(phi)
        (phi * w).sum().backward()
        g_heat, g_phi = a[1].grad.clone(), a[3].grad.clone()
        a[1].grad = a[3].grad = None
        rphi, rk, _, rm, rms, rmt = _piecewise(*a)
        (rphi * w).sum().backward()
        assert torch.equal(phi, rphi) and torch.equal(k, rk) and torch.equal(m, rm) and torch.equal(ms, rms)
        assert torch.isfinite(g_heat).all() and torch.isfinite(g_phi).all()
        assert torch.equal(g_phi, a[3].grad)
        close(g_heat, a[1].grad, atol=1e-5 * float(a[1].grad.abs().max()), rtol=1e-4)


def test_transporter_forward_uses_the_fused_kernel():
    """TransporterNet.forward ('max') goes through kp_keypoint_bottleneck_fwd: same 8-tuple as composing extract() by hand."""
    from keypoints_amd.models import functional as MF, transporter
    torch.manual_seed(1)
    net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 4).cuda()
    g = torch.Generator().manual_seed(2)
    xs = (torch.rand(3, 1, 84, 84, generator=g) * 2 - 1).cuda()
    xt = (torch.rand(3, 1, 84, 84, generator=g) * 2 - 1).cuda()
    net.eval()   # eval-mode BatchNorm: the two passes below see the same statistics
    with torch.no_grad():
        out = net(xs, xt)
        phi_s, _, _, _, m_s = net.extract(xs)
        phi_t, heat_t, k_t, p_t, m_t = net.extract(xt)
        phi, mask_s, mask_t = MF.transport_max(phi_s, phi_t, m_s, m_t)
        x_t = net.decoder(phi)
    for got, ref in zip((out[0], out[1], out[2], out[3], out[4][0], out[4][1], out[5], out[6], out[7]),
                        (x_t, phi, k_t, m_t, p_t[0], p_t[1], heat_t, mask_s, mask_t)):
        assert torch.equal(got, ref)
