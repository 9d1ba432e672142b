"""Row kernel of the single-channel first layers with 16 / 32 filters (conv3_first_fwd_rows_k, kp_conv_first.hip: the Pong-class
encoders' first convolution, /root/reference/keypoints/models/vgg.py:30-33 with the ReplicationPad2d(1) border physically present in
the input), through kp_conv_fwd / kp_conv_fwd_pair: against a torch fp32 convolution of the same 16-bit-rounded operands, against
the kernel it replaces (KP_CFG_FIRST_ROWS = 0: the stored tensor must be bit-identical, the BatchNorm statistics equal up to the
order of the fp32 sums), and the two-half launch against two separate launches (bit-identical, statistics rows included)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

KP_CFG_FIRST_ROWS = 14
ULP = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}

SHAPES = [  # N, Hi, Wi, K, workgroup cap (0 = the default)
    (16, 88, 88, 16, 0),     # BASELINE config 2's first layer at 16 images (86 pixel rows: iterations of 32, 32, 22)
    (16, 88, 88, 32, 0),     # 16-pixel iterations, the last one of 6
    (16, 88, 88, 16, 3),     # 3 workgroups: 12 wavefronts walk 1376 rows (ragged), statistics over many rows per lane
    (8, 20, 22, 16, 0),      # rows shorter than one iteration
    (4, 12, 90, 32, 2),
    (4, 10, 200, 16, 0),     # more than 128 pixels per input row: two dwords per lane
    (4, 10, 200, 32, 2),
    (64, 10, 10, 16, 0),     # maps narrower than 16 pixels: four output rows per workgroup would be MORE rows than the ceil(M / 64) the
    (64, 10, 10, 32, 0),     # kp_conv_fwd contract gives the statistics buffer (128 > 64): the grid is capped at M / 64
    (16, 6, 18, 16, 0),
]


@pytest.fixture(autouse=True)
def _knob():
    from keypoints_amd._lib import call
    yield
    call('kp_config', KP_CFG_FIRST_ROWS, 1)


def _operands(N, Hi, Wi, K, dt):
    g = torch.Generator().manual_seed(N * 131 + Hi * 7 + Wi + K)
    x = torch.randn(N, Hi, Wi, 1, generator=g).to(dt)
    w = (torch.randn(K, 3, 3, 1, generator=g) * (2.0 / 9) ** 0.5).to(dt)
    bias = torch.randn(K, generator=g)
    return x, w, bias


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('N,Hi,Wi,K,cap', SHAPES)
def test_rows_kernel_forward_and_statistics(N, Hi, Wi, K, cap, dt):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    code = _lib.KP_F16 if dt == torch.float16 else _lib.KP_BF16
    x, w, bias = _operands(N, Hi, Wi, K, dt)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double()).permute(0, 2, 3, 1).contiguous()
    xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
    Ho, Wo = Hi - 2, Wi - 2
    st = stream_ptr()
    out = {}
    for knob in (cap if cap else 1, 0):
        call('kp_config', KP_CFG_FIRST_ROWS, knob)
        y = torch.full((N, Ho, Wo, K), float('nan'), device='cuda', dtype=dt)
        stats = torch.full(((N * Ho * Wo + 63) // 64, 2, K), float('nan'), device='cuda')
        call('kp_conv_fwd', ptr(xd), ptr(wd), ptr(bd), ptr(y), ptr(stats), N, Hi, Wi, 1, K, 3, code, st)
        rows = _lib.last_rows()
        assert rows == _lib.lib().kp_conv_fwd_stats_rows(N, Hi, Wi, 1, K, 3, code)
        torch.cuda.synchronize()
        out[knob] = (y.cpu(), stats[:rows].cpu(), rows, stats.cpu())
    new, old = out[cap if cap else 1], out[0]
    want_rows = min((N * Ho + 3) // 4, cap if cap else 768, N * Ho * Wo // 64)
    assert new[2] == want_rows, 'the row kernel did not take this shape'
    assert new[2] <= new[3].shape[0], 'more statistics rows than the kp_conv_fwd contract allots (ceil(M / 64))'
    assert torch.isnan(new[3][new[2]:]).all()   # nothing written behind the rows the call reported
    assert torch.isfinite(new[0].float()).all() and torch.isfinite(new[1]).all()
    # the tensor: the same fp32 accumulation as the kernel it replaces
    assert torch.equal(new[0].view(torch.int16), old[0].view(torch.int16))
    # ... and one unit in the last place of the storage type from the exact result
    tol = ULP[dt] * ref.abs() + 2e-6 * ref.abs().max()
    assert ((new[0].double() - ref).abs() <= tol).all()
    # statistics: fp32 sums of the unrounded accumulators over all pixels, against fp64 sums of the exact result
    s = new[1].double().sum(0)
    M = N * Ho * Wo
    rs, rq = ref.reshape(M, K).sum(0), (ref * ref).reshape(M, K).sum(0)
    assert ((s[0] - rs).abs() <= 1e-5 * ref.abs().reshape(M, K).sum(0) + 1e-3).all()
    assert ((s[1] - rq).abs() <= 1e-5 * rq + 1e-3).all()
    so = old[1].double().sum(0)
    assert ((s - so).abs() <= 2e-5 * torch.stack((ref.abs().reshape(M, K).sum(0), rq)) + 1e-3).all()


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('N,Hi,Wi,K', [(32, 88, 88, 16), (32, 88, 88, 32), (16, 20, 22, 16)])
def test_rows_kernel_two_halves_in_one_grid(N, Hi, Wi, K, dt):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    code = _lib.KP_F16 if dt == torch.float16 else _lib.KP_BF16
    x, w, bias = _operands(N, Hi, Wi, K, dt)
    xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
    Ho, Wo, Nh = Hi - 2, Wi - 2, N // 2
    st = stream_ptr()
    assert _lib.lib().kp_conv_pair_supported(N, Ho, Wo, 1, K, 1, code)
    rows = _lib.lib().kp_conv_fwd_stats_rows(Nh, Hi, Wi, 1, K, 3, code)
    assert rows == min((Nh * Ho + 3) // 4, 768)
    y2 = torch.full((N, Ho, Wo, K), float('nan'), device='cuda', dtype=dt)
    s2 = torch.full((2, rows, 2, K), float('nan'), device='cuda')
    call('kp_conv_fwd_pair', ptr(xd), ptr(wd), ptr(bd), ptr(y2), ptr(s2), N, Hi, Wi, 1, K, code, st)
    for h in range(2):
        y1 = torch.full((Nh, Ho, Wo, K), float('nan'), device='cuda', dtype=dt)
        s1 = torch.full((rows, 2, K), float('nan'), device='cuda')
        xh = xd[h * Nh:(h + 1) * Nh].contiguous()
        call('kp_conv_fwd', ptr(xh), ptr(wd), ptr(bd), ptr(y1), ptr(s1), Nh, Hi, Wi, 1, K, 3, code, st)
        assert _lib.last_rows() == rows
        torch.cuda.synchronize()
        assert torch.equal(y1.view(torch.int16), y2[h * Nh:(h + 1) * Nh].view(torch.int16))
        assert torch.equal(s1, s2[h])
