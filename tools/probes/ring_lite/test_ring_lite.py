"""The border ring of a matrix-core data gradient on its own stream (conv3x3_ring_lite_k, kp_conv_mfma.hip: one wavefront per
workgroup, no LDS, fragments straight from L2 -- it runs BESIDE the persistent interior kernel, kp_config_ptr(KP_CFG_RING_STREAM)):
kp_conv_dgrad and kp_conv_dgrad_bnred must give bit-identical gradients and BatchNorm-backward rows with and without it, and the
gradient itself is checked against a torch fp64 transposed convolution of the same 16-bit operands (autograd of
/root/reference/keypoints/models/vgg.py:30-33: ReplicationPad2d(1) + Conv2d(3x3); gx lives on the padded domain)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

KP_CFG_BRED_PATCH, KP_CFG_RING_STREAM = 9, 15
ULP = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}

SHAPES = [  # N, H, W, K (gradient channels), C (input channels)
    (2, 16, 16, 64, 64),
    (3, 32, 16, 128, 64),
    (2, 16, 48, 64, 128),
    (4, 32, 32, 256, 256),
    (5, 16, 16, 512, 256),    # ring pixels: 5 x 18 = 90 per row group -- a ragged last tile in every edge group
]


@pytest.fixture(autouse=True)
def _knobs():
    from keypoints_amd._lib import call
    yield
    call('kp_config_ptr', KP_CFG_RING_STREAM, None)
    call('kp_config', KP_CFG_BRED_PATCH, 1)


def _operands(N, H, W, K, C, dt):
    g = torch.Generator().manual_seed(N * 31 + H + W * 7 + K + C)
    dy = torch.zeros(N, H + 4, W + 4, K, dtype=dt)
    dy[:, 2:-2, 2:-2] = (torch.randn(N, H, W, K, generator=g) * 0.5).to(dt)
    w = (torch.randn(K, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dt)       # forward layout [K][3][3][C]
    wd = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()                                # the data gradient's pack [C][3][3][K], flipped
    return dy, w, wd, g


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('N,H,W,K,C', SHAPES)
def test_data_gradient_with_the_ring_on_its_own_stream(N, H, W, K, C, dt):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    code = _lib.KP_F16 if dt == torch.float16 else _lib.KP_BF16
    dy, w, wd, _ = _operands(N, H, W, K, C, dt)
    # fp64: gradient of the VALID convolution of the padded input [H+2][W+2] -> output [H][W]
    ref = F.conv_transpose2d(dy[:, 2:-2, 2:-2].double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    dyd, wdd = dy.cuda(), wd.cuda()
    st = stream_ptr()
    side = torch.cuda.Stream()
    outs = []
    for ring in (None, side.cuda_stream):
        call('kp_config_ptr', KP_CFG_RING_STREAM, ring)
        gx = torch.full((N, H + 2, W + 2, C), float('nan'), device='cuda', dtype=dt)
        call('kp_conv_dgrad', ptr(dyd), ptr(wdd), ptr(gx), N, H, W, K, C, code, st)
        outs.append(gx.clone())        # (on torch's current stream: the call has made it wait for the ring)
    call('kp_config_ptr', KP_CFG_RING_STREAM, None)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[1].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    got = outs[1].double().cpu()
    tol = ULP[dt] * ref.abs() + 2e-5 * ref.abs().max()
    assert ((got - ref).abs() <= tol).all()


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('N,H,W,K,C', SHAPES)
def test_fused_batchnorm_rows_with_the_ring_on_its_own_stream(N, H, W, K, C, dt):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    code = _lib.KP_F16 if dt == torch.float16 else _lib.KP_BF16
    call('kp_config', KP_CFG_BRED_PATCH, 2)      # (128-channel output tiles too: every shape of the list takes the fused form)
    if _lib.lib().kp_conv_dgrad_bnred_supported(N, H, W, K, C, code) != 1:
        pytest.skip('no fused form for this shape')
    dy, w, wd, g = _operands(N, H, W, K, C, dt)
    yred = torch.randn(N, H, W, C, generator=g).to(dt)
    mean, invstd = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    dyd, wdd, yd = dy.cuda(), wd.cuda(), yred.cuda()
    md, isd, gd, bd = mean.cuda(), invstd.cuda(), gamma.cuda(), beta.cuda()
    rows = _lib.lib().kp_conv_dgrad_bnred_rows(N, H, W, K, C, code)
    st = stream_ptr()
    side = torch.cuda.Stream()
    outs = []
    for ring in (None, side.cuda_stream):
        call('kp_config_ptr', KP_CFG_RING_STREAM, ring)
        gx = torch.full((N, H + 2, W + 2, C), float('nan'), device='cuda', dtype=dt)
        part = torch.full((rows, 2, C), float('nan'), device='cuda')
        call('kp_conv_dgrad_bnred', ptr(dyd), ptr(wdd), ptr(gx), ptr(yd), ptr(md), ptr(isd), ptr(gd), ptr(bd), 0.0, ptr(part),
             N, H, W, K, C, code, st)
        assert _lib.last_rows() == rows
        outs.append((gx.clone(), part.clone()))
    call('kp_config_ptr', KP_CFG_RING_STREAM, None)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[1][1]).all()
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    assert torch.equal(outs[0][1], outs[1][1])
    # the rows against fp64 sums over the stored gradient, folded onto the unpadded map (the adjoint of the replicate pad)
    gxf = outs[1][0].double().cpu()
    yv = yred.double()
    xh = (yv - mean.double()) * invstd.double()
    z = xh * gamma.double() + beta.double()
    idx_h = torch.clamp(torch.arange(H + 2) - 1, 0, H - 1)
    idx_w = torch.clamp(torch.arange(W + 2) - 1, 0, W - 1)
    f = (z > 0).double()[:, idx_h][:, :, idx_w]
    gh = gxf * f
    want = torch.stack((gh.sum((0, 1, 2)), (gh * xh[:, idx_h][:, :, idx_w]).sum((0, 1, 2))))
    got = outs[1][1].double().sum(0).cpu()
    scale = torch.stack((gh.abs().sum((0, 1, 2)), (gh * xh[:, idx_h][:, :, idx_w]).abs().sum((0, 1, 2))))
    assert ((got - want).abs() <= 2e-5 * scale + 1e-4).all()


def test_unit_backward_is_bit_identical_with_the_ring_beside():
    """runtime.set_ring_beside through Unit.backward (the switch is off by default: measured, no gain)."""
    from keypoints_amd import runtime
    from keypoints_amd.models import knn, vgg
    res = []
    for on in (True, False):
        torch.manual_seed(4)
        unit = knn.Unit(64, 3, vgg.make_layers([64, 128, 'M', 128, 'U', 64])).cuda()
        unit.precision = 'fp16'
        x = torch.randn(4, 64, 32, 32, device='cuda', requires_grad=True)
        w = torch.randn(4, 3, 32, 32, device='cuda')
        runtime.set_ring_beside(on)
        try:
            out = unit(x)
            (out * w).sum().backward()
            torch.cuda.synchronize()
        finally:
            runtime.set_ring_beside(False)
        res.append((out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in unit.parameters()]))
    (o1, x1, p1), (o0, x0, p0) = res
    assert torch.equal(o1, o0) and torch.equal(x1, x0)
    for a, b in zip(p1, p0):
        assert torch.equal(a, b)
