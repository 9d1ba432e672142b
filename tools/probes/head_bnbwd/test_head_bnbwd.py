"""The head's backward fused with the BatchNorm backward of the 3x3 layer in front of it (kp_head_bnbwd_reduce / _apply,
include/kp_hip.h; vgg.py:30-38 -> knn.py:123-125 and their autograd) against the plain sequence it replaces -- kp_head_bwd_bnin,
kp_bn_act_bwd_reduce, kp_bn_bwd_finalize, kp_bn_act_bwd_apply -- through the C ABI, and at Unit level against the switch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('N,H,W,C,K,dp', [(8, 128, 128, 64, 3, 2), (16, 84, 84, 16, 1, 2), (4, 84, 84, 32, 4, 2), (3, 9, 13, 16, 2, 0), (2, 5, 7, 64, 3, 2)])
def test_fused_head_batchnorm_backward_equals_the_plain_sequence(N, H, W, C, K, dp, dt):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import precision_dtype, stream_ptr
    L = _lib.lib()
    tdt, code = precision_dtype(dt)
    st = stream_ptr()
    g = torch.Generator().manual_seed(N * 7 + C + K)
    y = torch.randn(N, H, W, C, generator=g).cuda().to(tdt)                       # the 3x3 layer's conv output
    mean, invstd = (torch.randn(C, generator=g) * 0.2).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    w = (torch.randn(K, C, generator=g) * 0.3).cuda()
    out = torch.randn(N, K, H, W, generator=g).cuda()                             # the head's saved output (sign matters only)
    gout = (torch.randn(N, K, H, W, generator=g) * 0.1).cuda()
    bn = (ptr(mean), ptr(invstd), ptr(gamma), ptr(beta))
    slope_in, slope = 0.0 if C == 64 else 0.01, 0.01
    rows = L.kp_head_bnbwd_rows(N, H, W, C, K, code)
    assert rows > 0
    # ---- plain sequence
    gx = torch.full((N, H, W, C), float('nan'), device='cuda', dtype=tdt)
    dw0, db0 = torch.zeros(K, C, device='cuda'), torch.zeros(K, device='cuda')
    ws0 = torch.empty(L.kp_head_bwd_ws_floats(N, H, W, C, K, code), device='cuda')
    call('kp_head_bwd_bnin', ptr(gout), ptr(out), ptr(y), *bn, slope_in, ptr(w), slope, ptr(gx), ptr(dw0), ptr(db0), ptr(ws0), N, H, W, C, K, code, st)
    part0 = torch.empty(N * H, 2, C, device='cuda')
    call('kp_bn_act_bwd_reduce', ptr(gx), 0, 0, None, ptr(y), *bn, slope_in, ptr(part0), N, H, W, C, code, st)
    prow0 = L.kp_bn_bwd_rows(0, 0, N, H, W, C, code)
    sums0, dg0, dbe0 = torch.empty(2 * C, device='cuda'), torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    wsd = torch.empty(128 * C, device='cuda', dtype=torch.float64)
    call('kp_bn_bwd_finalize', ptr(part0), ptr(wsd), prow0, C, ptr(sums0), ptr(dg0), ptr(dbe0), st)
    dy0 = torch.full((N, H + 2 * dp, W + 2 * dp, C), float('nan'), device='cuda', dtype=tdt)
    call('kp_bn_act_bwd_apply', ptr(gx), 0, 0, None, ptr(y), *bn, slope_in, ptr(sums0), ptr(dy0), dp, N, H, W, C, code, st)
    # ---- fused
    dw1, db1 = torch.zeros(K, C, device='cuda'), torch.zeros(K, device='cuda')
    ws1 = torch.empty(rows * (K * C + K), device='cuda')
    part1 = torch.full((rows, 2, C), float('nan'), device='cuda')
    hf = (ptr(gout), ptr(out), ptr(y), *bn, slope_in, ptr(w), slope)
    call('kp_head_bnbwd_reduce', *hf, ptr(dw1), ptr(db1), ptr(ws1), ptr(part1), N, H, W, C, K, code, st)
    sums1, dg1, dbe1 = torch.empty(2 * C, device='cuda'), torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    call('kp_bn_bwd_finalize', ptr(part1), ptr(wsd), rows, C, ptr(sums1), ptr(dg1), ptr(dbe1), st)
    dy1 = torch.full((N, H + 2 * dp, W + 2 * dp, C), float('nan'), device='cuda', dtype=tdt)
    call('kp_head_bnbwd_apply', *hf, ptr(sums0), ptr(dy1), dp, N, H, W, C, K, code, st)       # (the SAME sums: isolates the kernel)
    torch.cuda.synchronize()
    wscale = max(1e-6, float(dw0.abs().max()))
    assert float((dw1 - dw0).abs().max()) <= 1e-5 * wscale and float((db1 - db0).abs().max()) <= 1e-5 * max(1e-6, float(db0.abs().max()))
    assert float((sums1 - sums0).abs().max()) <= 2e-5 * float(sums0.abs().max())
    assert torch.isfinite(dy1.float()).all()
    assert torch.equal(dy1, dy0)                          # same sums -> the same gradient bit for bit, zero border included


@pytest.mark.parametrize('precision,typ,cin,z,K,size', [('fp16', 'F', 3, 64, 10, 128), ('bf16', 'VGG_PONG_LAYERNECK', 1, 16, 4, 84)])
def test_train_step_with_and_without_the_fused_head_backward(precision, typ, cin, z, K, size):
    """One Transporter step with kp_head_bnbwd_* (opt-in: set_fuse_head_bn_bwd(True)) and with the plain sequence (default): loss and outputs identical (forward untouched),
    the head's gradients bit-identical, every other gradient the same direction and size (the BatchNorm sums differ in their last
    bits: summation order)."""
    import keypoints_amd as ka
    from keypoints_amd.models import transporter
    g = torch.Generator().manual_seed(4)
    B = 8
    xs = torch.rand(B, cin, size, size, generator=g).cuda()
    xt = torch.rand(B, cin, size, size, generator=g).cuda()
    res = []
    for on in (True, False):
        ka.set_fuse_head_bn_bwd(on)
        try:
            torch.manual_seed(21)
            net = transporter.make(typ, cin, z, K).cuda().set_precision(precision)
            opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
            opt.zero_grad()
            out = net(xs, xt)
            loss = ka.l2_reconstruction_loss(out[0], xt)
            loss.backward()
            torch.cuda.synchronize()
            res.append((float(loss), out[0].detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}))
        finally:
            ka.set_fuse_head_bn_bwd(False)
    (l1, x1, g1), (l0, x0, g0) = res
    assert l1 == l0 and torch.equal(x1, x0)
    assert torch.equal(g1['decoder.out_block.0.weight'], g0['decoder.out_block.0.weight'])
    assert torch.equal(g1['decoder.out_block.0.bias'], g0['decoder.out_block.0.bias'])
    cos_min = 0.97 if precision == 'fp16' else 0.9
    for n in g0:
        a_, b_ = g1[n].double().flatten(), g0[n].double().flatten()
        if float(b_.abs().max()) == 0.0:
            assert float(a_.abs().max()) == 0.0, n
            continue
        cos = float((a_ * b_).sum() / (a_.norm() * b_.norm()))
        assert cos >= cos_min and 0.9 <= float(a_.norm() / b_.norm()) <= 1.1, (n, cos)
