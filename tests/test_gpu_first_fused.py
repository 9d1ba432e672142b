"""The recompute form of a Unit's image-sized first layer (kp_conv_first_stats / _fwd_bnact / _bwd_reduce / _bwd_wgrad,
include/kp_hip.h; knn.py:115-119 = ReplicationPad2d(1) + Conv2d(Cin <= 3 -> 64, 3x3) + BatchNorm2d (train) + LeakyReLU and their
autograd) against the plain entry points it replaces and against fp64 arithmetic.

What must hold: with the SAME BatchNorm constants the forward is bit-identical to kp_conv_fwd + kp_bn_act_fwd (the conv output is
rounded to the storage type where the plain path stores it, the activation is the same expression); statistics, backward sums and
the weight gradient are sums over another fixed partition: equal to fp32 summation order (asserted at 2e-5 / 1e-5 of the tensor's
scale, and against an fp64 evaluation of the same 16-bit values)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {'bf16': torch.bfloat16, 'fp16': torch.float16}


def _setup(N, C, H, W, dt, seed):
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import precision_dtype, stream_ptr
    tdt, code = precision_dtype(dt)
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(N, C, H, W, generator=g).cuda()
    w32 = (torch.randn(64, C, 3, 3, generator=g) * (2.0 / (64 * 9)) ** 0.5 * 3).cuda()
    bias = (torch.randn(64, generator=g) * 0.1).cuda()
    gamma = (torch.rand(64, generator=g) + 0.5).cuda()
    beta = (torch.randn(64, generator=g) * 0.2).cuda()
    st = stream_ptr()
    a = torch.empty(N, H + 2, W + 2, C, device='cuda', dtype=tdt)
    call('kp_nchw_to_nhwc', ptr(img), ptr(a), N, C, C, H, W, 1, code, st)
    wf = torch.empty(64, 3, 3, C, device='cuda', dtype=tdt)
    call('kp_pack_weights', ptr(w32), ptr(wf), None, 64, C, C, 3, code, st)
    return dict(N=N, C=C, H=H, W=W, tdt=tdt, code=code, st=st, a=a, wf=wf, bias=bias, gamma=gamma, beta=beta, gen=g)


def _finalize(stats, rows, count, K=64):
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    mean, invstd = torch.empty(K, device='cuda'), torch.empty(K, device='cuda')
    ws = torch.empty(128 * K, device='cuda', dtype=torch.float64)
    call('kp_bn_finalize', ptr(stats), rows, ptr(ws), ptr(mean), ptr(invstd), None, None, None, 0.1, 1e-5, count, K, stream_ptr())
    return mean, invstd


SHAPES = [(3, 3, 128, 128), (2, 1, 32, 64), (2, 2, 3, 32), (1, 3, 1, 32), (130, 3, 16, 32)]


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('N,C,H,W', SHAPES)
def test_fused_first_layer_forward(N, C, H, W, dt):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    L = _lib.lib()
    s = _setup(N, C, H, W, dt, 7 * N + C + H)
    code, st, K = s['code'], s['st'], 64
    assert L.kp_conv_first_fused_supported(N, H, W, C, K, code) == 1
    # plain path: convolution with statistics, finalize, BatchNorm + activation pass
    y = torch.empty(N, H, W, K, device='cuda', dtype=s['tdt'])
    rows0 = L.kp_conv_fwd_stats_rows(N, H + 2, W + 2, C, K, 3, code)
    stats0 = torch.zeros((N * H * W + 63) // 64, 2, K, device='cuda')
    call('kp_conv_fwd', ptr(s['a']), ptr(s['wf']), ptr(s['bias']), ptr(y), ptr(stats0), N, H + 2, W + 2, C, K, 3, code, st)
    mean0, inv0 = _finalize(stats0, rows0, N * H * W)
    a0 = torch.empty(N, H + 2, W + 2, K, device='cuda', dtype=s['tdt'])
    call('kp_bn_act_fwd', ptr(y), ptr(mean0), ptr(inv0), ptr(s['gamma']), ptr(s['beta']), 0.01, 0, ptr(a0), 1, None, N, H, W, K, code, st)
    # recompute path
    rows1 = L.kp_conv_first_fused_rows(N, H, W, C, K, code)
    assert 0 < rows1 <= 1024
    stats1 = torch.full((rows1, 2, K), float('nan'), device='cuda')
    call('kp_conv_first_stats', ptr(s['a']), ptr(s['wf']), ptr(s['bias']), ptr(stats1), N, H, W, C, K, code, st)
    mean1, inv1 = _finalize(stats1, rows1, N * H * W)
    assert float((mean1 - mean0).abs().max()) <= 2e-6 * float(mean0.abs().max()) + 1e-7
    assert float((inv1 / inv0 - 1).abs().max()) <= 2e-5      # (var = E[y^2] - E[y]^2: the cancellation amplifies the sums' last bits)
    # fp64 check of the statistics on the fp32 conv of the 16-bit operands
    xin = s['a'].float().permute(0, 3, 1, 2).double()
    yref = F.conv2d(xin, s['wf'].float().permute(0, 3, 1, 2).double(), s['bias'].double())
    assert float((mean1.double() - yref.mean(dim=(0, 2, 3))).abs().max()) <= 1e-5 * float(yref.abs().max())
    a1 = torch.full((N, H + 2, W + 2, K), float('nan'), device='cuda', dtype=s['tdt'])
    call('kp_conv_first_fwd_bnact', ptr(s['a']), ptr(s['wf']), ptr(s['bias']), ptr(mean0), ptr(inv0), ptr(s['gamma']), ptr(s['beta']), 0.01,
         ptr(a1), N, H, W, C, K, code, st)
    torch.cuda.synchronize()
    assert torch.isfinite(a1.float()).all()
    if N * H * W >= 4096 and N * H * W % 64 == 0:
        assert torch.equal(a1, a0)                  # same constants -> bit for bit, replicate border included
    else:   # (tiny shapes: the plain call runs another convolution kernel -- fp32 sums in another order before the one rounding)
        ulp = 2.0 ** (-7 if dt == 'bf16' else -10)
        assert float((a1.float() - a0.float()).abs().max()) <= ulp * max(1.0, float(a0.float().abs().max()))
        assert float((a1.float() != a0.float()).float().mean()) <= 0.02


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('N,C,H,W', SHAPES)
def test_fused_first_layer_backward(N, C, H, W, dt):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    L = _lib.lib()
    s = _setup(N, C, H, W, dt, 11 * N + C + W)
    code, st, K, tdt = s['code'], s['st'], 64, s['tdt']
    y = torch.empty(N, H, W, K, device='cuda', dtype=tdt)
    rows0 = L.kp_conv_fwd_stats_rows(N, H + 2, W + 2, C, K, 3, code)
    stats0 = torch.zeros((N * H * W + 63) // 64, 2, K, device='cuda')
    call('kp_conv_fwd', ptr(s['a']), ptr(s['wf']), ptr(s['bias']), ptr(y), ptr(stats0), N, H + 2, W + 2, C, K, 3, code, st)
    mean, inv = _finalize(stats0, rows0, N * H * W)
    g = (torch.randn(N, H + 2, W + 2, K, generator=s['gen']) * 0.05).cuda().to(tdt)     # dL/da over the padded domain
    bn = (ptr(mean), ptr(inv), ptr(s['gamma']), ptr(s['beta']))
    # plain path: reduce, finalize, apply (no data gradient: dp = 0), weight gradient, unpack
    part0 = torch.empty(N * H, 2, K, device='cuda')
    call('kp_bn_act_bwd_reduce', ptr(g), 1, 0, None, ptr(y), *bn, 0.01, ptr(part0), N, H, W, K, code, st)
    prow0 = L.kp_bn_bwd_rows(1, 0, N, H, W, K, code)
    sums0, dg0, db0 = torch.empty(2 * K, device='cuda'), torch.zeros(K, device='cuda'), torch.zeros(K, device='cuda')
    ws = torch.empty(128 * K, device='cuda', dtype=torch.float64)
    call('kp_bn_bwd_finalize', ptr(part0), ptr(ws), prow0, K, ptr(sums0), ptr(dg0), ptr(db0), st)
    dy = torch.empty(N, H, W, K, device='cuda', dtype=tdt)
    call('kp_bn_act_bwd_apply', ptr(g), 1, 0, None, ptr(y), *bn, 0.01, ptr(sums0), ptr(dy), 0, N, H, W, K, code, st)
    parts = L.kp_conv_wgrad_partials(N, H, W, C, K, 3, 0, code)
    sc0 = (torch.zeros if parts == 1 else torch.empty)(parts, K, 3, 3, C, device='cuda')
    call('kp_conv_wgrad', ptr(s['a']), ptr(dy), ptr(sc0), N, H, W, C, K, 3, 0, code, st)
    dw0 = torch.zeros(K, C, 3, 3, device='cuda')
    call('kp_wgrad_unpack', ptr(sc0), parts, ptr(dw0), K, C, C, 3, st)
    # recompute path
    rows1 = L.kp_conv_first_fused_rows(N, H, W, C, K, code)
    fargs = (ptr(g), ptr(s['a']), ptr(s['wf']), ptr(s['bias']), *bn, 0.01)
    part1 = torch.full((rows1, 2, K), float('nan'), device='cuda')
    call('kp_conv_first_bwd_reduce', *fargs, ptr(part1), N, H, W, C, K, code, st)
    sums1, dg1, db1 = torch.empty(2 * K, device='cuda'), torch.zeros(K, device='cuda'), torch.zeros(K, device='cuda')
    call('kp_bn_bwd_finalize', ptr(part1), ptr(ws), rows1, K, ptr(sums1), ptr(dg1), ptr(db1), st)
    scale = float(sums0.abs().max())
    assert float((sums1 - sums0).abs().max()) <= 2e-5 * scale, float((sums1 - sums0).abs().max()) / scale
    sc1 = torch.full((rows1, K, 3, 3, C), float('nan'), device='cuda')
    call('kp_conv_first_bwd_wgrad', *fargs, ptr(sums0), ptr(sc1), N, H, W, C, K, code, st)     # (the SAME sums: isolates the kernel)
    dw1 = torch.zeros(K, C, 3, 3, device='cuda')
    call('kp_wgrad_unpack', ptr(sc1), rows1, ptr(dw1), K, C, C, 3, st)
    torch.cuda.synchronize()
    assert torch.isfinite(dw1).all()
    wscale = float(dw0.abs().max())
    assert float((dw1 - dw0).abs().max()) <= 2e-5 * wscale, float((dw1 - dw0).abs().max()) / wscale
    # independent leg: the weight gradient in fp64 from the stored dy and the 16-bit image
    xin = s['a'].float().permute(0, 3, 1, 2).double()
    patches = F.unfold(xin, 3).view(N, C * 9, H * W)                                   # [N][C*9][HW], (c, r, s) major
    dref = torch.einsum('nkp,ncp->kc', dy.float().permute(0, 3, 1, 2).reshape(N, K, H * W).double(), patches).view(K, C, 3, 3)
    assert float((dw1.double() - dref).abs().max()) <= 2e-5 * float(dref.abs().max())


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_model_step_with_and_without_the_fused_first_layer(precision):
    """One Transporter F train step with the recompute first layer (default) and with the plain sequence: every output and every
    gradient agrees to the storage type's rounding (the statistics differ in their last bits, which flips single roundings)."""
    import keypoints_amd as ka
    from keypoints_amd.models import transporter
    g = torch.Generator().manual_seed(4)
    xs = torch.rand(8, 3, 128, 128, generator=g).cuda()
    xt = torch.rand(8, 3, 128, 128, generator=g).cuda()
    res = []
    for on in (True, False):
        ka.set_fuse_first_layer(on)
        try:
            torch.manual_seed(21)
            net = transporter.make('F', 3, 64, 10).cuda().set_precision(precision)
            opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
            if precision == 'fp16':
                opt.scaler = ka.amp.LossScaler('cuda')
            opt.zero_grad()
            out = net(xs, xt)
            loss = ka.l2_reconstruction_loss(out[0], xt)
            with ka.amp.scale_loss(loss, opt) as scaled:
                scaled.backward()
            inv = 1.0 / opt.scaler.scale if opt.scaler is not None else 1.0
            grads = {n: p.grad.detach().clone() * inv for n, p in net.named_parameters()}
            bufs = {n: b.detach().clone() for n, b in net.named_buffers()}
            torch.cuda.synchronize()
            res.append((float(loss), out[0].detach().clone(), out[2].detach().clone(), grads, bufs))
        finally:
            ka.set_fuse_first_layer(True)
    (l1, x1, k1, g1, b1), (l0, x0, k0, g0, b0) = res
    ulp = 2.0 ** (-8 if precision == 'bf16' else -11)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    assert abs(l1 - l0) <= 8 * ulp * abs(l0), (l1, l0)
    assert rel(x1, x0) <= 24 * ulp and float((k1 - k0).abs().max()) <= 16 * ulp
    # gradients at random init pass through 25 BatchNorm layers in 16-bit storage: a statistic's last bit flips single roundings and
    # the network amplifies them (the same two builds of the plain path differ as much when only a summation order changes), so the
    # wiring check is the direction and size of every gradient tensor, not its bits: cosine >= 0.97 (bf16: 0.9), norms within 10 %
    cos_min = 0.97 if precision == 'fp16' else 0.9
    worst = (None, 1.0)
    for n in g0:
        a_, b_ = g1[n].double().flatten(), g0[n].double().flatten()
        if float(b_.abs().max()) == 0.0:      # conv bias in front of BatchNorm: exactly zero in both
            assert float(a_.abs().max()) == 0.0, n
            continue
        cos = float((a_ * b_).sum() / (a_.norm() * b_.norm()))
        if cos < worst[1]:
            worst = (n, cos)
        assert cos >= cos_min and 0.9 <= float(a_.norm() / b_.norm()) <= 1.1, (n, cos, float(a_.norm() / b_.norm()))
    print('worst gradient cosine, fused vs plain first layer:', worst)
    for n in ('feature.in_block.1.weight', 'keypoint.in_block.1.weight', 'feature.in_block.2.weight', 'feature.in_block.2.bias'):
        assert float(g1[n].abs().max()) > 0, n             # the fused backward did write them
    for n in b0:
        if b0[n].dtype.is_floating_point:
            assert float((b1[n] - b0[n]).abs().max()) <= 1e-4 * max(1.0, float(b0[n].abs().max())), n
        else:
            assert torch.equal(b1[n], b0[n]), n
