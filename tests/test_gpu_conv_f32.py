"""Strict-fp32 3x3 convolution on the matrix cores (kp_conv_f32.hip: v_mfma_f32_32x32x2_f32) against fp64 torch
convolutions of the same fp32 operands, through the C ABI.  Tolerance: fp32 accumulation-order noise (the products are
exact, the sums of 9*C terms are rounded in a different order than any reference) -- 2e-6 of the layer's scale; the generic
VALU kernel (KP_DISABLE_MFMA is process-wide, so it is reached here through a shape the matrix-core kernel declines)
meets the same bound."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

FWD = [  # N, Ho, Wo, C, K, stats
    (2, 16, 16, 64, 64, True), (1, 8, 24, 64, 128, True), (3, 18, 18, 128, 64, False), (1, 16, 16, 512, 512, True),
    (2, 9, 11, 16, 64, True),            # M = 198: tail tile, last statistics row partial
    (1, 7, 5, 32, 128, False), (2, 34, 34, 256, 256, False), (1, 33, 31, 48, 192, True),
]


@pytest.mark.parametrize('N,Ho,Wo,C,K,with_stats', FWD)
def test_conv_fwd_f32_mfma(N, Ho, Wo, C, K, with_stats):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N * 1000 + C + K)
    x = torch.randn(N, Ho + 2, Wo + 2, C, generator=g)
    w = torch.randn(K, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5
    bias = torch.randn(K, generator=g) if with_stats else None
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2),
                   bias.double() if bias is not None else None).permute(0, 2, 3, 1).contiguous()
    xd, wd = x.cuda(), w.cuda()
    bd = bias.cuda() if bias is not None else None
    y = torch.full((N, Ho, Wo, K), float('nan'), device='cuda')
    rows = int(_lib.lib().kp_conv_fwd_stats_rows(N, Ho + 2, Wo + 2, C, K, 3, _lib.KP_F32)) if with_stats else 0
    assert rows == ((N * Ho * Wo + 63) // 64 if with_stats else 0)
    stats = torch.full((rows, 2, K), float('nan'), device='cuda') if with_stats else None
    call('kp_conv_fwd', ptr(xd), ptr(wd), ptr(bd), ptr(y), ptr(stats), N, Ho + 2, Wo + 2, C, K, 3, _lib.KP_F32, stream_ptr())
    got = y.double().cpu()
    assert torch.isfinite(got).all()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-6 * scale * max(1.0, (9 * C / 576) ** 0.5)   # ~ sqrt(#terms) growth
    if with_stats:
        s = stats.double().sum(0).cpu()
        assert torch.isfinite(s).all()
        flat = ref.reshape(-1, K)
        assert float((s[0] - flat.sum(0)).abs().max()) <= 2e-6 * float(flat.abs().sum(0).max())
        assert float((s[1] - (flat * flat).sum(0)).abs().max()) <= 2e-6 * float((flat * flat).sum(0).max())


@pytest.mark.parametrize('N,H,W,K,C', [(2, 16, 16, 64, 64), (1, 20, 12, 128, 256), (1, 64, 64, 128, 64)])
def test_conv_dgrad_f32(N, H, W, K, C):
    """kp_conv_dgrad in fp32 = the same kernel over the zero-bordered gradient with the flipped pack."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(H + W + K)
    wt = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5       # nn.Conv2d layout
    dy = torch.randn(N, K, H, W, generator=g)
    # gradient of conv(pad-free x[N][C][H+2][W+2]) w.r.t. x
    ref = F.conv_transpose2d(dy.double(), wt.double()).permute(0, 2, 3, 1).contiguous()    # [N][H+2][W+2][C]
    wdev = wt.cuda()
    wf = torch.empty(K, 3, 3, C, device='cuda'); wd = torch.empty(C, 3, 3, K, device='cuda')
    st = stream_ptr()
    call('kp_pack_weights', ptr(wdev), ptr(wf), ptr(wd), K, C, C, 3, _lib.KP_F32, st)
    dyp = torch.zeros(N, H + 4, W + 4, K, device='cuda')
    dyp[:, 2:-2, 2:-2] = dy.permute(0, 2, 3, 1).cuda()
    gx = torch.full((N, H + 2, W + 2, C), float('nan'), device='cuda')
    call('kp_conv_dgrad', ptr(dyp), ptr(wd), ptr(gx), N, H, W, K, C, _lib.KP_F32, st)
    got = gx.double().cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize('N,Ho,Wo,C,K,dp', [(2, 16, 16, 64, 64, 2), (3, 8, 24, 128, 64, 0), (1, 32, 32, 64, 256, 2),
                                            (2, 9, 11, 128, 128, 2), (8, 64, 64, 256, 128, 2)])
def test_conv_wgrad_f32_mfma(N, Ho, Wo, C, K, dp):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N + Ho + C + K)
    x = torch.randn(N, Ho + 2, Wo + 2, C, generator=g)
    dy = torch.randn(N, Ho, Wo, K, generator=g)
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(False)
    wz = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xd, wz) * dy.double().permute(0, 3, 1, 2)).sum().backward()
    ref = wz.grad                                                                   # [K][C][3][3]
    dyp = torch.zeros(N, Ho + 2 * dp, Wo + 2 * dp, K, device='cuda')
    dyp[:, dp:dp + Ho, dp:dp + Wo] = dy.cuda()
    parts = int(_lib.lib().kp_conv_wgrad_partials(N, Ho, Wo, C, K, 3, dp, _lib.KP_F32))
    scratch = torch.zeros(parts, K, 3, 3, C, device='cuda') if parts == 1 else torch.full((parts, K, 3, 3, C), float('nan'), device='cuda')
    st = stream_ptr()
    call('kp_conv_wgrad', ptr(x.cuda()), ptr(dyp), ptr(scratch), N, Ho, Wo, C, K, 3, dp, _lib.KP_F32, st)
    dw = torch.zeros(K, C, 3, 3, device='cuda')
    call('kp_wgrad_unpack', ptr(scratch), parts, ptr(dw), K, C, C, 3, st)
    got = dw.double().cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 3e-6 * float(ref.abs().max()) * max(1.0, (N * Ho * Wo / 4096) ** 0.5)
    # determinism: a second launch gives the same bits
    scratch2 = torch.zeros_like(scratch) if parts == 1 else torch.full_like(scratch, float('nan'))
    call('kp_conv_wgrad', ptr(x.cuda()), ptr(dyp), ptr(scratch2), N, Ho, Wo, C, K, 3, dp, _lib.KP_F32, st)
    assert torch.equal(scratch, scratch2)
