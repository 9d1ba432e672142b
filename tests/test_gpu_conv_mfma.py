"""bf16 MFMA implicit-GEMM convolution kernels vs a torch fp32 CPU convolution of the same bf16-rounded
operands (tolerance: fp32 accumulation-order noise + one bf16 rounding of the output)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _exact_reference_conv():
    """The GPU reference convolutions run on torch's native im2col + GEMM path (MIOpen off): exact fp32 arithmetic, and
    no per-shape kernel search (the MIOpen search alone took 85 s for the 256 -> 128 @ 128x128 reference)."""
    with torch.backends.cudnn.flags(enabled=False):
        yield

SHAPES = [  # N, Ho, Wo, C, K, with_stats
    (2, 16, 16, 64, 64, True), (2, 8, 8, 128, 128, True), (1, 32, 32, 64, 256, True),
    (1, 8, 24, 64, 128, True),          # M = 192: masked half tile
    (3, 18, 18, 128, 64, False),        # dgrad-like: M % 64 != 0, no bias / stats
    (1, 16, 16, 512, 512, True), (2, 130, 130, 64, 128, False),
    # LDS-patch kernel: 16x16-tileable outputs, and 16x16 interior + ring (data-gradient domain)
    (2, 32, 32, 128, 256, True), (1, 64, 64, 256, 128, True), (3, 16, 16, 64, 512, True),
    (2, 34, 34, 256, 256, False), (1, 66, 66, 128, 64, False), (2, 18, 18, 512, 128, False),
]


@pytest.mark.parametrize('N,Ho,Wo,C,K,with_stats', SHAPES)
def test_conv_fwd_mfma(N, Ho, Wo, C, K, with_stats):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N * 1000 + C + K)
    x = torch.randn(N, Ho + 2, Wo + 2, C, generator=g).bfloat16()
    w = (torch.randn(K, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).bfloat16()
    bias = torch.randn(K, generator=g) if with_stats else None
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias).permute(0, 2, 3, 1).contiguous()
    xd, wd = x.cuda(), w.cuda()
    bd = bias.cuda() if bias is not None else None
    y = torch.full((N, Ho, Wo, K), float('nan'), device='cuda', dtype=torch.bfloat16)
    rows = (N * Ho * Wo + 63) // 64
    stats = torch.full((rows, 2, K), float('nan'), device='cuda') if with_stats else None
    st = stream_ptr()
    call('kp_prof_enable', 1)
    call('kp_conv_fwd', ptr(xd), ptr(wd), ptr(bd), ptr(y), ptr(stats), N, Ho + 2, Wo + 2, C, K, 3, _lib.KP_BF16, st)
    import ctypes
    n = ctypes.c_longlong()
    call('kp_prof_collect', 2, None, None, ctypes.addressof(n))
    call('kp_prof_enable', 0)
    assert n.value == 1, 'the MFMA kernel did not take this shape'
    got = y.float().cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    assert err <= 1e-2 * ref.abs().max().item() + 1e-3, err
    # tighter: compare with the bf16 rounding of the reference
    close = (got - ref.bfloat16().float()).abs() <= 2 ** -7 * ref.abs() + 1e-3
    assert close.float().mean().item() > 0.999
    if with_stats:
        s = stats[:_lib.last_rows()].double().sum(0).cpu()
        assert torch.isfinite(s).all()
        flat = ref.double().reshape(-1, K)
        assert (s[0] - flat.sum(0)).abs().max().item() <= 1e-3 * flat.abs().sum(0).max().item()
        assert (s[1] - (flat ** 2).sum(0)).abs().max().item() <= 1e-3 * (flat ** 2).sum(0).max().item()


WG_SHAPES = [  # N, Ho, Wo, C, K, dy_pad
    (2, 16, 16, 64, 64, 2), (2, 8, 8, 128, 128, 2), (1, 32, 32, 64, 128, 0), (2, 16, 16, 128, 64, 2),
    (4, 16, 16, 256, 512, 2), (1, 128, 128, 64, 128, 2),
    # tap-reuse kernel (kp_wgrad_patch.hip): (128 x 64) and (64 x 128) gradient blocks, odd split counts
    (3, 32, 32, 128, 128, 2), (2, 16, 32, 128, 64, 2), (1, 64, 64, 256, 64, 0), (5, 16, 16, 64, 512, 2),
    (2, 24, 48, 64, 128, 2),
    # per-tap fallback (kp_wgrad_mfma.hip: maps smaller than the tap-reuse kernel's blocks), several pixel splits
    (32, 8, 8, 512, 512, 2), (64, 4, 4, 512, 256, 2), (16, 8, 8, 64, 64, 0),
]


@pytest.mark.parametrize('N,Ho,Wo,C,K,dp', WG_SHAPES)
def test_conv_wgrad_mfma(N, Ho, Wo, C, K, dp):
    import ctypes
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N * 77 + C + K)
    x = torch.randn(N, Ho + 2, Wo + 2, C, generator=g).bfloat16()
    dyi = torch.randn(N, Ho, Wo, K, generator=g).bfloat16()
    dy = torch.zeros(N, Ho + 2 * dp, Wo + 2 * dp, K, dtype=torch.bfloat16)
    dy[:, dp:dp + Ho, dp:dp + Wo] = dyi
    # reference: dW[k][r][s][c] = sum dy[n,p,q,k] * x[n,p+r,q+s,c]
    xf, df = x.double(), dyi.double()
    ref = torch.zeros(K, 3, 3, C, dtype=torch.float64)
    for r in range(3):
        for s in range(3):
            ref[:, r, s, :] = torch.einsum('npqk,npqc->kc', df, xf[:, r:r + Ho, s:s + Wo])
    parts = _lib.lib().kp_conv_wgrad_partials(N, Ho, Wo, C, K, 3, dp, _lib.KP_BF16)
    scratch = torch.zeros(K, 3, 3, C, device='cuda') if parts == 1 else \
        torch.full((parts, K, 3, 3, C), float('nan'), device='cuda')   # partial buffers need no zeroing
    xd, dyd = x.cuda(), dy.cuda()   # keep alive: ptr() of a temporary would dangle
    call('kp_prof_enable', 1)
    call('kp_conv_wgrad', ptr(xd), ptr(dyd), ptr(scratch), N, Ho, Wo, C, K, 3, dp, _lib.KP_BF16, stream_ptr())
    n = ctypes.c_longlong()
    call('kp_prof_collect', 3, None, None, ctypes.addressof(n))
    call('kp_prof_enable', 0)
    assert n.value == 1, 'the MFMA wgrad kernel did not take this shape'
    got = scratch.double().cpu() if parts == 1 else scratch.double().sum(0).cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item() * (N * Ho * Wo) ** 0.5 + 1e-4, (err, ref.abs().max().item())
    assert ((got - ref).norm() / ref.norm()).item() < 1e-5


def test_wgrad_fallback_is_bitwise_repeatable():
    """The per-tap fallback kernel leaves one partial gradient per pixel split (no fp32 atomics any more): two runs give
    the same bits, every partial buffer is written completely."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    N, Ho, Wo, C, K, dp = 32, 8, 8, 512, 512, 2
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, Ho + 2, Wo + 2, C, generator=g).bfloat16().cuda()
    dy = torch.zeros(N, Ho + 2 * dp, Wo + 2 * dp, K, dtype=torch.bfloat16)
    dy[:, dp:dp + Ho, dp:dp + Wo] = torch.randn(N, Ho, Wo, K, generator=g).bfloat16()
    dy = dy.cuda()
    parts = _lib.lib().kp_conv_wgrad_partials(N, Ho, Wo, C, K, 3, dp, _lib.KP_BF16)
    assert parts > 1
    outs = []
    for _ in range(3):
        scratch = torch.full((parts, K, 3, 3, C), float('nan'), device='cuda')
        call('kp_conv_wgrad', ptr(x), ptr(dy), ptr(scratch), N, Ho, Wo, C, K, 3, dp, _lib.KP_BF16, stream_ptr())
        gw = torch.zeros(K, C, 3, 3, device="cuda")   # kp_wgrad_unpack ACCUMULATES into the gradient arena
        call('kp_wgrad_unpack', ptr(scratch), parts, ptr(gw), K, C, C, 3, stream_ptr())
        assert torch.isfinite(scratch).all()
        outs.append(gw)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


FIRST_SHAPES = [  # N, Ho, Wo, C, K
    (2, 32, 32, 3, 64), (1, 16, 16, 1, 16), (2, 128, 128, 3, 64), (3, 16, 48, 3, 32), (1, 8, 8, 4, 64),
    # K = 16 / 32: sliding-window weight gradient (one contiguous row segment per pixel stream), ragged segments, Wo < streams
    (4, 84, 84, 1, 16), (64, 20, 37, 1, 32), (64, 9, 5, 3, 16), (32, 33, 50, 4, 32), (32, 6, 3, 1, 16), (4, 84, 84, 1, 32),
    # MFMA first-layer kernels (K = 64, Cin <= 3, Wo % 32 == 0): forward from 4096 pixels, weight gradient from 65536;
    # partially filled and empty wavefronts (150 groups over 20 wavefronts), Cin = 1 and 2
    (16, 64, 64, 3, 64), (1, 50, 96, 3, 64), (7, 96, 160, 3, 64), (8, 128, 128, 1, 64), (6, 128, 96, 2, 64),
]


@pytest.mark.parametrize('N,Ho,Wo,C,K', FIRST_SHAPES)
def test_conv_first_layer(N, Ho, Wo, C, K):
    """Cin <= 4 first-layer kernels (kp_conv_first.hip): forward + BN partial sums + weight gradient."""
    import ctypes
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N * 31 + C * 7 + K)
    x = torch.randn(N, Ho + 2, Wo + 2, C, generator=g).bfloat16()
    w = (torch.randn(K, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).bfloat16()
    bias = torch.randn(K, generator=g)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias).permute(0, 2, 3, 1).contiguous()
    xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
    y = torch.full((N, Ho, Wo, K), float('nan'), device='cuda', dtype=torch.bfloat16)
    rows = N * Ho * Wo // 64
    stats = torch.full((rows, 2, K), float('nan'), device='cuda')
    st = stream_ptr()
    n = ctypes.c_longlong()
    call('kp_prof_enable', 1)
    call('kp_conv_fwd', ptr(xd), ptr(wd), ptr(bd), ptr(y), ptr(stats), N, Ho + 2, Wo + 2, C, K, 3, _lib.KP_BF16, st)
    call('kp_prof_collect', 4, None, None, ctypes.addressof(n))
    assert n.value == 1, 'the first-layer kernel did not take this shape'
    got = y.float().cpu()
    assert torch.isfinite(got).all()
    close = (got - ref.bfloat16().float()).abs() <= 2 ** -7 * ref.abs() + 1e-3
    assert close.float().mean().item() > 0.9995
    s = stats[:_lib.last_rows()].double().sum(0).cpu()
    flat = ref.double().reshape(-1, K)
    assert (s[0] - flat.sum(0)).abs().max().item() <= 1e-4 * flat.abs().sum(0).max().item()
    assert (s[1] - (flat ** 2).sum(0)).abs().max().item() <= 1e-4 * (flat ** 2).sum(0).max().item()

    for dp in (0, 2):
        dyi = torch.randn(N, Ho, Wo, K, generator=g).bfloat16()
        dy = torch.zeros(N, Ho + 2 * dp, Wo + 2 * dp, K, dtype=torch.bfloat16)
        dy[:, dp:dp + Ho, dp:dp + Wo] = dyi
        xf, df = x.double(), dyi.double()
        wref = torch.zeros(K, 3, 3, C, dtype=torch.float64)
        for r in range(3):
            for s_ in range(3):
                wref[:, r, s_, :] = torch.einsum('npqk,npqc->kc', df, xf[:, r:r + Ho, s_:s_ + Wo])
        parts = _lib.lib().kp_conv_wgrad_partials(N, Ho, Wo, C, K, 3, dp, _lib.KP_BF16)
        assert parts >= 1   # per-workgroup partials, no atomics (one workgroup when the whole problem is one band of rows)
        scratch = torch.zeros(K, 3, 3, C, device='cuda') if parts == 1 else \
            torch.full((parts, K, 3, 3, C), float('nan'), device='cuda')
        dyd = dy.cuda()
        call('kp_prof_enable', 1)
        call('kp_conv_wgrad', ptr(xd), ptr(dyd), ptr(scratch), N, Ho, Wo, C, K, 3, dp, _lib.KP_BF16, st)
        call('kp_prof_collect', 4, None, None, ctypes.addressof(n))
        assert n.value == 1, 'the first-layer wgrad kernel did not take this shape'
        gotw = scratch.double().cpu() if parts == 1 else scratch.double().sum(0).cpu()
        assert torch.isfinite(gotw).all()
        assert ((gotw - wref).norm() / wref.norm()).item() < 1e-5
    call('kp_prof_enable', 0)


@pytest.mark.parametrize('N,H,W,K,C', [(2, 16, 16, 64, 64), (3, 16, 32, 128, 256), (1, 64, 64, 128, 64),
                                       (2, 32, 32, 512, 128), (130, 16, 16, 64, 128), (1, 8, 8, 64, 64)])
def test_conv_dgrad(N, H, W, K, C):
    """kp_conv_dgrad (LDS-patch interior + 3-tap border ring) == valid correlation of the zero-bordered dy."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N * 13 + H + K + C)
    dy = torch.zeros(N, H + 4, W + 4, K, dtype=torch.bfloat16)
    dy[:, 2:2 + H, 2:2 + W] = torch.randn(N, H, W, K, generator=g).bfloat16()
    wd = (torch.randn(C, 3, 3, K, generator=g) * (2.0 / (9 * K)) ** 0.5).bfloat16()
    ref = F.conv2d(dy.float().permute(0, 3, 1, 2), wd.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    dyd, wdd = dy.cuda(), wd.cuda()
    gx = torch.full((N, H + 2, W + 2, C), float('nan'), device='cuda', dtype=torch.bfloat16)
    call('kp_conv_dgrad', ptr(dyd), ptr(wdd), ptr(gx), N, H, W, K, C, _lib.KP_BF16, stream_ptr())
    got = gx.float().cpu()
    assert torch.isfinite(got).all()
    close = (got - ref.bfloat16().float()).abs() <= 2 ** -7 * ref.abs() + 1e-3
    assert close.float().mean().item() > 0.999
    assert (got - ref).abs().max().item() <= 1e-2 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize('N,Ho,Wo,C,K', [(2, 20, 84, 16, 32), (3, 9, 33, 32, 32), (2, 84, 84, 32, 16), (1, 7, 5, 16, 16),
                                         (4, 32, 32, 16, 32)])
def test_conv_small_channels(N, Ho, Wo, C, K):
    """C, K in {16, 32} kernels (kp_conv_small.hip): forward + BN partial sums, data-gradient form, weight gradient."""
    import ctypes
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N * 3 + Ho + Wo + C * 5 + K)
    x = torch.randn(N, Ho + 2, Wo + 2, C, generator=g).bfloat16()
    w = (torch.randn(K, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).bfloat16()
    bias = torch.randn(K, generator=g)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias).permute(0, 2, 3, 1).contiguous()
    xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
    y = torch.full((N, Ho, Wo, K), float('nan'), device='cuda', dtype=torch.bfloat16)
    rows = (N * Ho * Wo + 63) // 64
    stats = torch.full((rows, 2, K), float('nan'), device='cuda')
    st = stream_ptr()
    n = ctypes.c_longlong()
    call('kp_prof_enable', 1)
    call('kp_conv_fwd', ptr(xd), ptr(wd), ptr(bd), ptr(y), ptr(stats), N, Ho + 2, Wo + 2, C, K, 3, _lib.KP_BF16, st)
    used = _lib.last_rows()
    call('kp_prof_collect', 5, None, None, ctypes.addressof(n))
    if N * Ho * Wo >= 256:   # tiny problems (fewer than 4 partial rows in the buffer) stay on the general kernel
        assert n.value == 1, 'the small-channel kernel did not take this shape'
    got = y.float().cpu()
    assert torch.isfinite(got).all()
    close = (got - ref.bfloat16().float()).abs() <= 2 ** -7 * ref.abs() + 1e-3
    assert close.float().mean().item() > 0.9995
    s = stats[:used].double().sum(0).cpu()
    flat = ref.double().reshape(-1, K)
    assert (s[0] - flat.sum(0)).abs().max().item() <= 1e-4 * flat.abs().sum(0).max().item()
    assert (s[1] - (flat ** 2).sum(0)).abs().max().item() <= 1e-4 * (flat ** 2).sum(0).max().item()
    # no bias / no stats (the data-gradient use)
    y2 = torch.full((N, Ho, Wo, K), float('nan'), device='cuda', dtype=torch.bfloat16)
    call('kp_conv_fwd', ptr(xd), ptr(wd), None, ptr(y2), None, N, Ho + 2, Wo + 2, C, K, 3, _lib.KP_BF16, st)
    ref2 = ref - bias
    close = (y2.float().cpu() - ref2.bfloat16().float()).abs() <= 2 ** -7 * ref2.abs() + 1e-3
    assert close.float().mean().item() > 0.9995
    for dp in (0, 2):
        dyi = torch.randn(N, Ho, Wo, K, generator=g).bfloat16()
        dy = torch.zeros(N, Ho + 2 * dp, Wo + 2 * dp, K, dtype=torch.bfloat16)
        dy[:, dp:dp + Ho, dp:dp + Wo] = dyi
        xf, df = x.double(), dyi.double()
        wref = torch.zeros(K, 3, 3, C, dtype=torch.float64)
        for r in range(3):
            for s_ in range(3):
                wref[:, r, s_, :] = torch.einsum('npqk,npqc->kc', df, xf[:, r:r + Ho, s_:s_ + Wo])
        parts = _lib.lib().kp_conv_wgrad_partials(N, Ho, Wo, C, K, 3, dp, _lib.KP_BF16)
        assert parts >= 1   # per-workgroup partials, no atomics (one workgroup when the whole problem is one band of rows)
        scratch = torch.zeros(K, 3, 3, C, device='cuda') if parts == 1 else \
            torch.full((parts, K, 3, 3, C), float('nan'), device='cuda')
        dyd = dy.cuda()
        call('kp_prof_enable', 1)
        call('kp_conv_wgrad', ptr(xd), ptr(dyd), ptr(scratch), N, Ho, Wo, C, K, 3, dp, _lib.KP_BF16, st)
        call('kp_prof_collect', 9, None, None, ctypes.addressof(n))     # KP_PROF_WGRAD_SMALL
        assert n.value == 1, 'the small-channel wgrad kernel did not take this shape'
        gotw = scratch.double().cpu() if parts == 1 else scratch.double().sum(0).cpu()
        assert torch.isfinite(gotw).all()
        assert ((gotw - wref).norm() / wref.norm()).item() < 1e-5
        scratch2 = torch.full_like(scratch, float('nan'))            # bitwise reproducible: same partials again
        call('kp_conv_wgrad', ptr(xd), ptr(dyd), ptr(scratch2), N, Ho, Wo, C, K, 3, dp, _lib.KP_BF16, st)
        assert torch.equal(scratch, scratch2)
    call('kp_prof_enable', 0)


@pytest.mark.parametrize('C,K,H', [(256, 128, 128), (512, 512, 32)])
def test_conv_full_size_against_gpu_fp32(C, K, H):
    """BASELINE-size layers (128 images): forward (+ BN sums), data-gradient and weight-gradient kernels against an fp32
    convolution of the same bf16-rounded operands computed on the GPU by torch (the CPU oracle takes minutes here)."""
    import ctypes
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    N = 128
    g = torch.Generator(device='cuda').manual_seed(C + K + H)
    x = torch.randn(N, H + 2, H + 2, C, device='cuda', generator=g).bfloat16()
    w = (torch.randn(K, 3, 3, C, device='cuda', generator=g) * (2.0 / (9 * C)) ** 0.5).bfloat16()
    bias = torch.randn(K, device='cuda', generator=g)
    st = stream_ptr()
    # every output buffer starts as NaN: a tile the persistent scheduler skips, or writes twice wrongly, cannot pass
    y = torch.full((N, H, H, K), float('nan'), device='cuda', dtype=torch.bfloat16)
    stats = torch.full((N * H * H // 64, 2, K), float('nan'), device='cuda')
    call('kp_conv_fwd', ptr(x), ptr(w), ptr(bias), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3, _lib.KP_BF16, st)
    rows = _lib.last_rows()
    xf, wf = x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2)
    ref = F.conv2d(xf, wf, bias).permute(0, 2, 3, 1)
    got = y.float()
    assert torch.isfinite(got).all()
    # ALL elements within one bf16 ulp (2^-7 relative) of the fp32 result + fp32 summation-order noise; 99.95 % within
    # half an ulp (the rest are sums that land on the other side of a rounding boundary)
    err = (got - ref).abs()
    assert float((err - (2 ** -7 * ref.abs() + 2e-4)).max()) <= 0, float(err.max())
    assert (err <= 2 ** -8 * ref.abs() + 2e-4).float().mean().item() > 0.9995
    # BatchNorm sums per channel: one missing 16x16 tile shifts sum(y^2) by ~256 * E[y^2] = 1.2e-4 of the total
    s = stats[:rows].double().sum(0)
    assert torch.isfinite(s).all()
    flat = ref.double().reshape(-1, K)
    assert float(((s[0] - flat.sum(0)).abs() / flat.abs().sum(0)).max()) <= 2e-5
    assert float(((s[1] - (flat ** 2).sum(0)).abs() / (flat ** 2).sum(0)).max()) <= 2e-5
    del ref, got, flat, err
    # backward: dy with a zero border of 2
    dyi = torch.randn(N, H, H, K, device='cuda', generator=g).bfloat16()
    dy = torch.zeros(N, H + 4, H + 4, K, device='cuda', dtype=torch.bfloat16)
    dy[:, 2:-2, 2:-2] = dyi
    wd = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()            # [C][3][3][K], rotated by 180 degrees
    gx = torch.full((N, H + 2, H + 2, C), float('nan'), device='cuda', dtype=torch.bfloat16)
    call('kp_conv_dgrad', ptr(dy), ptr(wd), ptr(gx), N, H, H, K, C, _lib.KP_BF16, st)
    # gradient over the PADDED input domain = full correlation of dy with the rotated weights
    gref = torch.nn.grad.conv2d_input((N, C, H + 2, H + 2), wf, dyi.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    gg = gx.float()
    assert torch.isfinite(gg).all()
    err = (gg - gref).abs()
    assert float((err - (2 ** -7 * gref.abs() + 2e-4)).max()) <= 0, float(err.max())
    assert (err <= 2 ** -8 * gref.abs() + 2e-4).float().mean().item() > 0.9995
    del gref, gg, err
    parts = _lib.lib().kp_conv_wgrad_partials(N, H, H, C, K, 3, 2, _lib.KP_BF16)
    scratch = torch.zeros(parts, K, 3, 3, C, device='cuda') if parts == 1 else \
        torch.full((parts, K, 3, 3, C), float('nan'), device='cuda')
    call('kp_conv_wgrad', ptr(x), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 2, _lib.KP_BF16, st)
    dw = torch.zeros(K, C, 3, 3, device='cuda')
    call('kp_wgrad_unpack', ptr(scratch), parts, ptr(dw), K, C, C, 3, st)
    wref = torch.nn.grad.conv2d_weight(xf, wf.shape, dyi.float().permute(0, 3, 1, 2))
    assert torch.isfinite(dw).all()
    assert ((dw - wref).norm() / wref.norm()).item() < 2e-4
    # every entry: a pixel block dropped from one (k, c) pair moves that entry by ~1/sqrt(blocks) of its size
    assert float((dw - wref).abs().max()) <= 2e-3 * float(wref.abs().max())


def test_first_layer_full_size_against_gpu_fp32():
    """The image-sized first layer (3 -> 64 at 128x128, 128 images) on its MFMA kernels against torch's fp32 GPU convolution."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    N, H, C, K = 128, 128, 3, 64
    g = torch.Generator(device='cuda').manual_seed(77)
    x = torch.randn(N, H + 2, H + 2, C, device='cuda', generator=g).bfloat16()
    w = (torch.randn(K, 3, 3, C, device='cuda', generator=g) * (2.0 / 27) ** 0.5).bfloat16()
    bias = torch.randn(K, device='cuda', generator=g)
    st = stream_ptr()
    y = torch.empty(N, H, H, K, device='cuda', dtype=torch.bfloat16)
    stats = torch.empty(N * H * H // 64, 2, K, device='cuda')
    call('kp_conv_fwd', ptr(x), ptr(w), ptr(bias), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3, _lib.KP_BF16, st)
    rows = _lib.last_rows()
    xf, wf = x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2)
    ref = F.conv2d(xf, wf, bias).permute(0, 2, 3, 1)
    assert ((y.float() - ref).abs() <= 2 ** -7 * ref.abs() + 1e-3).float().mean().item() > 0.9995
    s = stats[:rows].double().sum(0)
    flat = ref.double().reshape(-1, K)
    assert (s[0] - flat.sum(0)).abs().max().item() <= 1e-4 * flat.abs().sum(0).max().item()
    assert (s[1] - (flat ** 2).sum(0)).abs().max().item() <= 1e-4 * (flat ** 2).sum(0).max().item()
    dyi = torch.randn(N, H, H, K, device='cuda', generator=g).bfloat16()
    dy = torch.zeros(N, H + 4, H + 4, K, device='cuda', dtype=torch.bfloat16)
    dy[:, 2:-2, 2:-2] = dyi
    parts = _lib.lib().kp_conv_wgrad_partials(N, H, H, C, K, 3, 2, _lib.KP_BF16)
    assert parts > 1
    scratch = torch.full((parts, K, 3, 3, C), float('nan'), device='cuda')
    call('kp_conv_wgrad', ptr(x), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 2, _lib.KP_BF16, st)
    dw = torch.zeros(K, C, 3, 3, device='cuda')
    call('kp_wgrad_unpack', ptr(scratch), parts, ptr(dw), K, C, C, 3, st)
    wref = torch.nn.grad.conv2d_weight(xf, wf.shape, dyi.float().permute(0, 3, 1, 2))
    assert ((dw - wref).norm() / wref.norm()).item() < 1e-4
    # bitwise reproducible: a second run gives the same partials
    scratch2 = torch.full((parts, K, 3, 3, C), float('nan'), device='cuda')
    call('kp_conv_wgrad', ptr(x), ptr(dy), ptr(scratch2), N, H, H, C, K, 3, 2, _lib.KP_BF16, st)
    assert torch.equal(scratch, scratch2)


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('N,H,W,C,K,dp', [(4, 20, 28, 64, 128, 2), (3, 12, 12, 128, 128, 2), (2, 50, 70, 128, 64, 0), (5, 9, 17, 64, 128, 2)])
def test_weight_gradient_on_zero_padded_copies_off_the_pixel_block(N, H, W, C, K, dp, dt):
    """Matrix-core channels on a map that is not a multiple of the tap-reuse kernel's 8 x 16 pixel block (no BASELINE shape):
    kp_conv_wgrad_pad_hw offers the padded size, kp_pad_hw makes the zero-padded copies, and the tap-reuse kernel's gradient equals
    the per-tap fallback's on the original tensors (fp32 summation order) and an fp64 evaluation of the same 16-bit operands."""
    import ctypes
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import precision_dtype, stream_ptr
    L = _lib.lib()
    tdt, code = precision_dtype(dt)
    st = stream_ptr()
    g = torch.Generator().manual_seed(N * 31 + H)
    a = (torch.randn(N, H + 2, W + 2, C, generator=g) * 0.5).cuda().to(tdt)
    dy = torch.zeros(N, H + 2 * dp, W + 2 * dp, K, device='cuda', dtype=tdt)
    dy[:, dp:dp + H, dp:dp + W] = (torch.randn(N, H, W, K, generator=g) * 0.5).cuda().to(tdt)
    hp, wp = ctypes.c_int(0), ctypes.c_int(0)
    assert L.kp_conv_wgrad_pad_hw(N, H, W, C, K, dp, code, ctypes.byref(hp), ctypes.byref(wp)) == 1
    Hp, Wp = hp.value, wp.value
    assert Hp % 8 == 0 and Wp % 16 == 0 and 0 <= Hp - H < 8 and 0 <= Wp - W < 16
    assert L.kp_conv_wgrad_pad_hw(N, Hp, Wp, C, K, dp, code, None, None) == 0          # already a multiple: nothing to pad
    xp = torch.full((N, Hp + 2, Wp + 2, C), float('nan'), device='cuda', dtype=tdt)
    dyp = torch.full((N, Hp + 2 * dp, Wp + 2 * dp, K), float('nan'), device='cuda', dtype=tdt)
    call('kp_pad_hw', ptr(a), ptr(xp), N, H + 2, W + 2, Hp + 2, Wp + 2, C, code, st)
    call('kp_pad_hw', ptr(dy), ptr(dyp), N, H + 2 * dp, W + 2 * dp, Hp + 2 * dp, Wp + 2 * dp, K, code, st)
    assert torch.equal(xp[:, :H + 2, :W + 2], a) and int((xp != 0).sum()) == int((a != 0).sum())
    assert torch.equal(dyp[:, :H + 2 * dp, :W + 2 * dp], dy) and int((dyp != 0).sum()) == int((dy != 0).sum())   # zeros everywhere else

    def wgrad(x_, dy_, h_, w_):
        parts = L.kp_conv_wgrad_partials(N, h_, w_, C, K, 3, dp, code)
        sc = (torch.zeros if parts == 1 else torch.empty)(parts, K, 3, 3, C, device='cuda')
        call('kp_conv_wgrad', ptr(x_), ptr(dy_), ptr(sc), N, h_, w_, C, K, 3, dp, code, st)
        dw = torch.zeros(K, C, 3, 3, device='cuda')
        call('kp_wgrad_unpack', ptr(sc), parts, ptr(dw), K, C, C, 3, st)
        return dw
    dw_pad, dw_plain = wgrad(xp, dyp, Hp, Wp), wgrad(a, dy, H, W)
    torch.cuda.synchronize()
    scale = float(dw_plain.abs().max())
    assert float((dw_pad - dw_plain).abs().max()) <= 2e-5 * scale
    import torch.nn.functional as F
    patches = F.unfold(a.float().permute(0, 3, 1, 2).double(), 3).view(N, C * 9, H * W)
    d = dy[:, dp:dp + H, dp:dp + W].float().permute(0, 3, 1, 2).reshape(N, K, H * W).double()
    ref = torch.einsum('nkp,ncp->kc', d, patches).view(K, C, 3, 3)
    assert float((dw_pad.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
