"""Fast bf16 BatchNorm/activation kernels (kp_bn_fast.hip) against the general kernels (kp_bn.hip) on the same
inputs (A/B through kp_config): identical arithmetic, so results must agree to fp32 summation-order noise."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(2, 16, 16, 64), (3, 8, 8, 512), (1, 12, 20, 128), (2, 32, 32, 256), (5, 6, 10, 64)]


def _params(C, g):
    mean = torch.randn(C, generator=g).cuda()
    invstd = (torch.rand(C, generator=g) + 0.5).cuda()
    gamma = (torch.randn(C, generator=g) * 0.5 + 1).cuda()
    beta = torch.randn(C, generator=g).cuda()
    return mean, invstd, gamma, beta


@pytest.mark.parametrize('N,H,W,C', SHAPES)
@pytest.mark.parametrize('pad', [0, 1])
@pytest.mark.parametrize('tail', ['pool', 'up'])
def test_bn_act_fwd_tail_fast(N, H, W, C, pad, tail):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N * 100 + C + pad + 5 * (tail == 'up'))
    y = torch.randn(N, H, W, C, generator=g).bfloat16().cuda()
    mean, invstd, gamma, beta = _params(C, g)
    tcode = _lib.TAIL_MAXPOOL2 if tail == 'pool' else _lib.TAIL_UPSAMPLE2
    Ho, Wo = (H // 2, W // 2) if tail == 'pool' else (2 * H, 2 * W)
    res = []
    for fast in (0, 1):
        call('kp_config', 4, fast)
        out = torch.full((N, Ho + 2 * pad, Wo + 2 * pad, C), float('nan'), device='cuda', dtype=torch.bfloat16)
        am = torch.full((N, Ho, Wo, C), 9, device='cuda', dtype=torch.uint8) if tail == 'pool' else None
        call('kp_bn_act_fwd', ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), 0.01, tcode, ptr(out), pad,
             ptr(am), N, H, W, C, _lib.KP_BF16, stream_ptr())
        res.append((out.float().cpu(), am.cpu() if am is not None else None))
    call('kp_config', 4, 1)
    assert torch.isfinite(res[1][0]).all()
    assert torch.equal(res[0][0], res[1][0])
    if tail == 'pool':
        assert torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize('N,H,W,C', SHAPES)
@pytest.mark.parametrize('pad', [0, 1])
def test_bn_act_fwd_fast(N, H, W, C, pad):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N * 100 + C + pad)
    y = torch.randn(N, H, W, C, generator=g).bfloat16().cuda()
    mean, invstd, gamma, beta = _params(C, g)
    outs = []
    for fast in (0, 1):
        call('kp_config', 4, fast)
        out = torch.full((N, H + 2 * pad, W + 2 * pad, C), float('nan'), device='cuda', dtype=torch.bfloat16)
        call('kp_bn_act_fwd', ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), 0.01, _lib.TAIL_NONE, ptr(out), pad,
             None, N, H, W, C, _lib.KP_BF16, stream_ptr())
        outs.append(out.float().cpu())
    call('kp_config', 4, 1)
    assert torch.isfinite(outs[1]).all()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('N,H,W,C', SHAPES)
@pytest.mark.parametrize('tail', ['none', 'pool'])
@pytest.mark.parametrize('gp', [0, 1])
def test_bn_bwd_fast(N, H, W, C, tail, gp):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N * 100 + C + gp * 7 + (tail == 'pool'))
    tcode = _lib.TAIL_MAXPOOL2 if tail == 'pool' else _lib.TAIL_NONE
    Hs, Ws = (H // 2, W // 2) if tail == 'pool' else (H, W)
    y = torch.randn(N, H, W, C, generator=g).bfloat16().cuda()
    gsrc = torch.randn(N, Hs + 2 * gp, Ws + 2 * gp, C, generator=g).bfloat16().cuda()
    argmax = torch.randint(0, 4, (N, Hs, Ws, C), generator=g, dtype=torch.uint8).cuda() if tail == 'pool' else None
    mean, invstd, gamma, beta = _params(C, g)
    st = stream_ptr()
    res = []
    for fast in (0, 1):
        call('kp_config', 4, fast)
        partials = torch.full((N * H, 2, C), float('nan'), device='cuda')
        call('kp_bn_act_bwd_reduce', ptr(gsrc), gp, tcode, ptr(argmax), ptr(y), ptr(mean), ptr(invstd), ptr(gamma),
             ptr(beta), 0.01, ptr(partials), N, H, W, C, _lib.KP_BF16, st)
        tot = partials[:_lib.last_rows()].double().sum(0).cpu()
        sums = tot.float().reshape(-1).cuda()
        outs = []
        for dp in (0, 2):
            dy = torch.full((N, H + 2 * dp, W + 2 * dp, C), float('nan'), device='cuda', dtype=torch.bfloat16)
            call('kp_bn_act_bwd_apply', ptr(gsrc), gp, tcode, ptr(argmax), ptr(y), ptr(mean), ptr(invstd), ptr(gamma),
                 ptr(beta), 0.01, ptr(sums), ptr(dy), dp, N, H, W, C, _lib.KP_BF16, st)
            outs.append(dy.float().cpu())
        res.append((tot, outs))
    call('kp_config', 4, 1)
    (t0, o0), (t1, o1) = res
    assert torch.isfinite(t1).all()
    scale = t0.abs().max().item() + 1.0
    assert (t0 - t1).abs().max().item() <= 1e-4 * scale
    for a, b in zip(o0, o1):
        assert torch.isfinite(b).all()
        # sums differ by summation-order noise -> a few results may round to the neighbouring bf16
        close = (a - b).abs() <= 2 ** -7 * a.abs() + 1e-3
        assert close.float().mean().item() > 0.9999


@pytest.mark.parametrize('N,H,W,C', SHAPES + [(2, 64, 64, 128), (1, 7, 128, 64)])
@pytest.mark.parametrize('gp', [0, 1])
def test_upsample2_bwd_fast(N, H, W, C, gp):
    """Separable adjoint of the bilinear x2 tail (with the replicate-pad border folded in) vs the general gather
    kernel and vs torch autograd of F.interpolate(align_corners=True) + F.pad(replicate) in fp32."""
    import torch.nn.functional as F
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N * 100 + C + gp)
    gsrc = torch.randn(N, 2 * H + 2 * gp, 2 * W + 2 * gp, C, generator=g).bfloat16().cuda()
    outs = []
    for fast in (0, 1):
        call('kp_config', 4, fast)
        out = torch.full((N, H, W, C), float('nan'), device='cuda', dtype=torch.bfloat16)
        call('kp_upsample2_bwd', ptr(gsrc), gp, ptr(out), N, H, W, C, _lib.KP_BF16, stream_ptr())
        outs.append(out.float().cpu())
    call('kp_config', 4, 1)
    a, b = outs
    assert torch.isfinite(b).all()
    x = torch.zeros(N, C, H, W, requires_grad=True)
    up = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
    if gp:
        up = F.pad(up, (gp,) * 4, mode='replicate')
    up.backward(gsrc.float().cpu().permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1)
    for got in (a, b):   # one bf16 rounding of the result
        assert ((got - ref).abs() <= 2 ** -7 * ref.abs() + 1e-3).float().mean().item() > 0.9999
    assert ((a - b).abs() <= 2 ** -7 * a.abs() + 1e-3).float().mean().item() > 0.9999
