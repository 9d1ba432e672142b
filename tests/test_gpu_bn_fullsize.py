"""BatchNorm(train) + activation (+ max-pool / bilinear-upsample tail, + replicate pad) kernels at the bench's layer
shapes (128 images) against an INDEPENDENT reference: torch.nn.functional.batch_norm / leaky_relu / max_pool2d /
interpolate / pad and their autograd, in fp32 on the GPU, fed the same bf16-rounded tensors.

Forward: every element within one bf16 ulp of the fp32 reference (outputs are NaN-prefilled: an unwritten or wrong
tile fails), 99.9 % within half an ulp.  Backward: dgamma / dbeta to 1e-4 of their scale, dy within one bf16 ulp plus
the bf16 resolution of the gradient's scale.  The statistics handed to the kernels are those of kp_bn_finalize on the
kernel's own partial sums' format (here: computed in fp64 from the same data)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(128, 128, 128, 64), (128, 64, 64, 256), (128, 16, 16, 512)]
ULP = 2.0 ** -7      # bf16: 8 significant bits -> 1 ulp <= 2^-7 relative
EPS = 1e-5


def _ref_forward(y32, gamma, beta, slope, tail, pad):
    """y32: [N,C,H,W] fp32 leaf.  Returns the padded NCHW activation (fp32) as torch computes it."""
    z = F.batch_norm(y32, None, None, gamma, beta, training=True, momentum=0.0, eps=EPS)
    a = F.leaky_relu(z, slope) if slope > 0 else F.relu(z)
    if tail == 'pool':
        a = F.max_pool2d(a, 2, 2)
    elif tail == 'up':
        a = F.interpolate(a, scale_factor=2, mode='bilinear', align_corners=True)
    if pad:
        a = F.pad(a, (pad,) * 4, mode='replicate')
    return a


def _within(got, ref, abs_tol):
    err = (got - ref).abs()
    bound = ULP * ref.abs() + abs_tol
    return float((err - bound).max()), float((err <= 0.5 * ULP * ref.abs() + abs_tol).float().mean())


@pytest.mark.parametrize('N,H,W,C', SHAPES)
@pytest.mark.parametrize('tail,slope', [('none', 0.01), ('pool', 0.01), ('up', 0.0)])
def test_bn_act_full_size_vs_torch(N, H, W, C, tail, slope):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, last_rows, ptr
    from keypoints_amd.runtime import stream_ptr
    if tail == 'up' and H == 128:
        pytest.skip('no upsample follows a 128x128 layer in any BASELINE net (the result would be 256x256x128 images)')
    st = stream_ptr()
    g = torch.Generator(device='cuda').manual_seed(H + C)
    # conv-like pre-activations with per-channel offset and scale
    y = (torch.randn(N, H, W, C, device='cuda', generator=g) * (0.5 + torch.rand(C, device='cuda', generator=g))
         + torch.randn(C, device='cuda', generator=g)).bfloat16()
    gamma = torch.rand(C, device='cuda', generator=g) + 0.5
    beta = torch.randn(C, device='cuda', generator=g) * 0.3
    yd = y.double()
    mean64 = yd.mean(dim=(0, 1, 2))
    var64 = yd.var(dim=(0, 1, 2), unbiased=False)
    mean, invstd = mean64.float(), torch.rsqrt(var64 + EPS).float()
    tcode = {'none': _lib.TAIL_NONE, 'pool': _lib.TAIL_MAXPOOL2, 'up': _lib.TAIL_UPSAMPLE2}[tail]
    Ho, Wo = {'none': (H, W), 'pool': (H // 2, W // 2), 'up': (2 * H, 2 * W)}[tail]
    pad = 1

    # ---- forward ------------------------------------------------------------------------------------------
    out = torch.full((N, Ho + 2, Wo + 2, C), float('nan'), device='cuda', dtype=torch.bfloat16)
    am = torch.full((N, Ho, Wo, C), 9, device='cuda', dtype=torch.uint8) if tail == 'pool' else None
    call('kp_bn_act_fwd', ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), slope, tcode, ptr(out), pad, ptr(am),
         N, H, W, C, _lib.KP_BF16, st)
    y32 = y.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    gam = gamma.clone().requires_grad_(True)
    bet = beta.clone().requires_grad_(True)
    ref = _ref_forward(y32, gam, bet, slope, tail, pad)
    got = out.float().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    worst, frac = _within(got, ref.detach(), 2e-5)
    assert worst <= 0 and frac > 0.999, (worst, frac)
    if tail == 'pool':
        assert int(am.max()) <= 3
    del got

    # ---- backward -----------------------------------------------------------------------------------------
    gsrc = torch.randn(N, Ho + 2, Wo + 2, C, device='cuda', generator=g).bfloat16()
    ref.backward(gsrc.float().permute(0, 3, 1, 2))
    del ref
    gq, gp = gsrc, 1
    btail = tcode
    if tail == 'up':
        g2 = torch.full((N, H, W, C), float('nan'), device='cuda', dtype=torch.bfloat16)
        call('kp_upsample2_bwd', ptr(gsrc), 1, ptr(g2), N, H, W, C, _lib.KP_BF16, st)
        gq, gp, btail = g2, 0, _lib.TAIL_NONE
    partials = torch.full((N * H, 2, C), float('nan'), device='cuda')
    call('kp_bn_act_bwd_reduce', ptr(gq), gp, btail, ptr(am), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
         slope, ptr(partials), N, H, W, C, _lib.KP_BF16, st)
    rows = last_rows()
    sums = torch.full((2 * C,), float('nan'), device='cuda')
    dgam = torch.zeros(C, device='cuda')
    dbet = torch.zeros(C, device='cuda')
    ws = torch.empty(128 * C, device='cuda', dtype=torch.float64)
    call('kp_bn_bwd_finalize', ptr(partials), ptr(ws), rows, C, ptr(sums), ptr(dgam), ptr(dbet), st)
    # the upsample adjoint hands a bf16-rounded gradient to the reduction: the sums carry that rounding -- random, 2^-9
    # relative per element, and like the sums themselves it grows as sqrt(N*H*W): measured 1.5e-3 of the scale
    for got_v, ref_v, name in ((dbet, bet.grad, 'dbeta'), (dgam, gam.grad, 'dgamma')):
        scale = float(ref_v.abs().max())
        assert float((got_v - ref_v).abs().max()) <= (5e-3 if tail == 'up' else 1e-4) * scale, name
    for dp in (2, 0):
        dy = torch.full((N, H + 2 * dp, W + 2 * dp, C), float('nan'), device='cuda', dtype=torch.bfloat16)
        call('kp_bn_act_bwd_apply', ptr(gq), gp, btail, ptr(am), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
             slope, ptr(sums), ptr(dy), dp, N, H, W, C, _lib.KP_BF16, st)
        d = dy.float()
        assert torch.isfinite(d).all()
        if dp:
            assert float(d[:, :dp].abs().max()) == 0 and float(d[:, -dp:].abs().max()) == 0
            assert float(d[:, :, :dp].abs().max()) == 0 and float(d[:, :, -dp:].abs().max()) == 0
            d = d[:, dp:-dp, dp:-dp]
        refd = y32.grad.permute(0, 2, 3, 1)
        scale = float(refd.abs().max())
        # one bf16 rounding of dy (two half-ulp roundings for 'up': the upsample adjoint is stored in bf16 first)
        err = (d - refd).abs()
        assert float((err - (1.01 * ULP * refd.abs() + 1e-4 * scale)).max()) <= 0, (dp, float(err.max()), scale)
        assert float((err <= 0.5 * ULP * refd.abs() + 1e-4 * scale).float().mean()) > (0.9 if tail == 'up' else 0.999)
        del err
        del d, dy
