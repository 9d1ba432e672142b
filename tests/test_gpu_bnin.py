"""BatchNorm-apply + activation + replicate pad of a 3x3 layer fused into the NEXT 3x3 layer's kernels (kp_conv_fwd_bnin,
kp_conv_wgrad_bnin; vgg.py:30-38 layer pairs of the Pong-class nets, vgg.py:53,66) against the separate pass it replaces
(kp_bn_act_fwd + kp_conv_fwd / kp_conv_wgrad): the same expression and the same rounding to the storage type feed the same
MFMAs, so every output bit must agree -- on NaN-prefilled outputs, at ragged sizes, with and without BatchNorm, and through a
whole train step of the Pong transporter."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(3, 84, 84), (2, 37, 50), (4, 5, 7), (1, 1, 3), (2, 32, 32), (5, 16, 33)]


def _bn(C, g, on=True):
    if not on:
        return (None,) * 4
    return ((torch.randn(C, generator=g) * 0.3).cuda(), (torch.rand(C, generator=g) + 0.5).cuda(),
            (torch.randn(C, generator=g) * 0.5 + 1).cuda(), (torch.randn(C, generator=g) * 0.3).cuda())


@pytest.mark.parametrize('N,H,W', SHAPES)
@pytest.mark.parametrize('C,K', [(16, 16), (16, 32), (32, 16), (32, 32)])
@pytest.mark.parametrize('dt,bn_on,slope', [('bf16', True, 0.01), ('fp16', True, 0.0), ('bf16', False, 0.01)])
def test_conv_fwd_and_wgrad_bnin_equal_the_separate_pass(N, H, W, C, K, dt, bn_on, slope):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    tdt, code = (torch.bfloat16, _lib.KP_BF16) if dt == 'bf16' else (torch.float16, _lib.KP_F16)
    L = _lib.lib()
    if L.kp_conv_bnin_supported(N, H, W, C, K, code) != 1:
        pytest.skip('no fused kernel for this shape (the engine then runs the separate pass)')
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + C + K)
    yprev = torch.randn(N, H, W, C, generator=g).to(tdt).cuda()
    bn = _bn(C, g, bn_on)
    w = (torch.randn(K, 3, 3, C, generator=g) * 0.1).to(tdt).cuda()
    bias = (torch.randn(K, generator=g) * 0.1).cuda()
    st = stream_ptr()
    # reference: the separate pass, then the plain kernels
    a = torch.full((N, H + 2, W + 2, C), float('nan'), device='cuda', dtype=tdt)
    call('kp_bn_act_fwd', ptr(yprev), *[ptr(t) for t in bn], slope, _lib.TAIL_NONE, ptr(a), 1, None, N, H, W, C, code, st)
    rows = L.kp_conv_fwd_stats_rows(N, H + 2, W + 2, C, K, 3, code)
    nrows = (N * H * W + 63) // 64
    outs = []
    for fused in (False, True):
        y = torch.full((N, H, W, K), float('nan'), device='cuda', dtype=tdt)
        stats = torch.full((nrows, 2, K), float('nan'), device='cuda', dtype=torch.float32)
        if fused:
            call('kp_conv_fwd_bnin', ptr(yprev), *[ptr(t) for t in bn], slope, ptr(w), ptr(bias), ptr(y), ptr(stats), N, H, W, C, K, code, st)
        else:
            call('kp_conv_fwd', ptr(a), ptr(w), ptr(bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, code, st)
        outs.append((y.float().cpu(), stats[:rows].cpu()))
    assert torch.isfinite(outs[1][0]).all() and torch.isfinite(outs[1][1]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # sanity of the reference itself against torch (the fused path is then right by the equality above)
    af = a.float().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(af, w.float().permute(0, 3, 1, 2), bias).permute(0, 2, 3, 1).cpu()
    # every element within ONE ulp of the storage type of the fp32 convolution of the same stored activation (+ fp32 accumulation-order noise)
    ulp = 2.0 ** (-7 if dt == 'bf16' else -10)
    assert ((outs[1][0] - ref).abs() <= ulp * ref.abs() + 2e-5 * max(1.0, float(ref.abs().max()))).all()
    # weight gradient
    for dp in (0, 2):
        dy = torch.zeros(N, H + 2 * dp, W + 2 * dp, K, dtype=tdt)
        dy[:, dp:dp + H, dp:dp + W] = (torch.randn(N, H, W, K, generator=g) * 0.1).to(tdt)
        dy = dy.cuda()
        parts = L.kp_conv_wgrad_partials(N, H, W, C, K, 3, dp, code)
        res = []
        for fused in (False, True):
            scratch = (torch.zeros if parts == 1 else lambda *s, **k: torch.full(s, float('nan'), **k))(parts, K, 3, 3, C, device='cuda', dtype=torch.float32)
            if fused:
                call('kp_conv_wgrad_bnin', ptr(yprev), *[ptr(t) for t in bn], slope, ptr(dy), ptr(scratch), N, H, W, C, K, dp, code, st)
            else:
                call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, W, C, K, 3, dp, code, st)
            res.append(scratch.cpu())
        assert torch.isfinite(res[1]).all() and torch.equal(res[0], res[1])


@pytest.mark.parametrize('N,H,W', [(3, 84, 84), (2, 37, 50), (4, 8, 8), (1, 1, 3), (2, 32, 32), (5, 16, 33)])
@pytest.mark.parametrize('C,K', [(32, 4), (16, 1), (32, 16), (64, 3), (16, 10), (128, 16)])
@pytest.mark.parametrize('dt,bn_on,slope_in', [('bf16', True, 0.01), ('fp16', True, 0.0), ('bf16', False, 0.01)])
def test_head_fwd_and_bwd_bnin_equal_the_separate_pass(N, H, W, C, K, dt, bn_on, slope_in):
    """The 1x1 head fed by the last 3x3 layer's conv output (kp_head_fwd_bnin / kp_head_bwd_bnin; knn.py:110-130) against
    kp_bn_act_fwd + kp_head_fwd / kp_head_bwd: output, input gradient, weight and bias gradient bit for bit."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    tdt, code = (torch.bfloat16, _lib.KP_BF16) if dt == 'bf16' else (torch.float16, _lib.KP_F16)
    L = _lib.lib()
    if L.kp_head_bnin_supported(N, H, W, C, K, code) != 1:
        pytest.skip('no fused head kernel for this shape (the engine then runs the separate pass)')
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + C + K)
    yprev = torch.randn(N, H, W, C, generator=g).to(tdt).cuda()
    bn = _bn(C, g, bn_on)
    w = (torch.randn(K, C, generator=g) * 0.2).cuda()
    bias = (torch.randn(K, generator=g) * 0.1).cuda()
    gout = torch.randn(N, K, H, W, generator=g).cuda()
    st = stream_ptr()
    a = torch.full((N, H, W, C), float('nan'), device='cuda', dtype=tdt)
    call('kp_bn_act_fwd', ptr(yprev), *[ptr(t) for t in bn], slope_in, _lib.TAIL_NONE, ptr(a), 0, None, N, H, W, C, code, st)
    res = []
    for fused in (False, True):
        out = torch.full((N, K, H, W), float('nan'), device='cuda')
        if fused:
            call('kp_head_fwd_bnin', ptr(yprev), *[ptr(t) for t in bn], slope_in, ptr(w), ptr(bias), 0.2, ptr(out), N, H, W, C, K, code, st)
        else:
            call('kp_head_fwd', ptr(a), ptr(w), ptr(bias), 0.2, ptr(out), N, H, W, C, K, code, st)
        dx = torch.full((N, H, W, C), float('nan'), device='cuda', dtype=tdt)
        dw, db = torch.zeros(K, C, device='cuda'), torch.zeros(K, device='cuda')
        ws = torch.full((L.kp_head_bwd_ws_floats(N, H, W, C, K, code),), float('nan'), device='cuda')
        if fused:
            call('kp_head_bwd_bnin', ptr(gout), ptr(out), ptr(yprev), *[ptr(t) for t in bn], slope_in, ptr(w), 0.2, ptr(dx), ptr(dw),
                 ptr(db), ptr(ws), N, H, W, C, K, code, st)
        else:
            call('kp_head_bwd', ptr(gout), ptr(out), ptr(a), ptr(w), 0.2, ptr(dx), ptr(dw), ptr(db), ptr(ws), N, H, W, C, K, code, st)
        res.append([t.float().cpu() for t in (out, dx, dw, db)])
    for r, f in zip(*res):
        assert torch.isfinite(f).all() and torch.equal(r, f)
    # sanity of the reference itself against torch
    ref = torch.nn.functional.leaky_relu(torch.einsum('nhwc,kc->nkhw', a.float(), w) + bias.view(1, K, 1, 1), 0.2).cpu()
    # fp32 output of fp32 weights x the stored activation (carried as a hi + lo pair of 16-bit values: 2^-17 relative)
    assert (res[1][0] - ref).abs().max() <= 3e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('N,H,W', [(3, 84, 84), (2, 37, 50), (4, 5, 7), (1, 1, 3), (2, 32, 32), (5, 16, 33)])
@pytest.mark.parametrize('K,C', [(16, 16), (32, 16), (16, 32), (32, 32)])
@pytest.mark.parametrize('dt,bn_on,slope', [('bf16', True, 0.01), ('fp16', True, 0.0), ('bf16', False, 0.01)])
def test_dgrad_with_the_batchnorm_backward_sums_in_its_epilogue(N, H, W, K, C, dt, bn_on, slope):
    """kp_conv_dgrad_bnred against kp_conv_dgrad + kp_bn_act_bwd_reduce(gp = 1): the padded-domain gradient bit for bit, the sums
    of the layer below (after kp_bn_bwd_finalize) to fp32 summation accuracy -- the fused kernel adds them in another order and in
    the padded domain (fold = adjoint of the replicate pad)."""
    _check_dgrad_bnred(N, H, W, K, C, dt, bn_on, slope)


@pytest.mark.parametrize('N,H,W,K,C,cus', [(4, 32, 32, 128, 256, 8), (6, 16, 16, 512, 64, 8), (2, 48, 48, 64, 128, 8), (8, 16, 32, 256, 128, 16),
                                           (3, 16, 16, 64, 64, 256), (4, 64, 64, 128, 64, 256), (8, 32, 32, 256, 512, 32)])
@pytest.mark.parametrize('dt,bn_on,slope', [('fp16', True, 0.01), ('bf16', True, 0.0), ('fp16', False, 0.01)])
def test_dgrad_bnred_of_the_matrix_core_layers(N, H, W, K, C, cus, dt, bn_on, slope):
    """The same contract for model F's layers (C, K multiples of 64, 16x16-tileable maps): the sums come from the epilogue of the
    persistent LDS-patch kernel (interior) and of the 3-tap ring kernel.  The CU budget shrinks the grid so that workgroups walk
    several tiles."""
    from keypoints_amd._lib import call
    call('kp_config', 6, cus)
    call('kp_config', 9, 2)      # KP_CFG_BRED_PATCH: the 128-channel tile form too (off by default: slower than the separate pass)
    try:
        _check_dgrad_bnred(N, H, W, K, C, dt, bn_on, slope)
    finally:
        call('kp_config', 6, 0)
        call('kp_config', 9, 1)


def _check_dgrad_bnred(N, H, W, K, C, dt, bn_on, slope):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    tdt, code = (torch.bfloat16, _lib.KP_BF16) if dt == 'bf16' else (torch.float16, _lib.KP_F16)
    L = _lib.lib()
    if L.kp_conv_dgrad_bnred_supported(N, H, W, K, C, code) != 1:
        pytest.skip('no fused kernel for this shape (the engine then runs the separate pass)')
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + C + K)
    dy = torch.zeros(N, H + 4, W + 4, K, dtype=tdt)
    dy[:, 2:2 + H, 2:2 + W] = (torch.randn(N, H, W, K, generator=g) * 0.1).to(tdt)
    dy = dy.cuda()
    wd = (torch.randn(C, 3, 3, K, generator=g) * 0.1).to(tdt).cuda()
    yred = torch.randn(N, H, W, C, generator=g).to(tdt).cuda()
    bn = _bn(C, g, bn_on)
    st = stream_ptr()
    nrows = (N * (H + 2) * (W + 2) + 63) // 64
    # reference: plain data gradient, then the separate reduce pass over the folded gradient
    gx0 = torch.full((N, H + 2, W + 2, C), float('nan'), device='cuda', dtype=tdt)
    call('kp_conv_dgrad', ptr(dy), ptr(wd), ptr(gx0), N, H, W, K, C, code, st)
    p0 = torch.full((N * H, 2, C), float('nan'), device='cuda')
    call('kp_bn_act_bwd_reduce', ptr(gx0), 1, _lib.TAIL_NONE, None, ptr(yred), *[ptr(t) for t in bn], slope, ptr(p0), N, H, W, C, code, st)
    r0 = L.kp_bn_bwd_rows(1, _lib.TAIL_NONE, N, H, W, C, code)
    gx1 = torch.full((N, H + 2, W + 2, C), float('nan'), device='cuda', dtype=tdt)
    r1 = L.kp_conv_dgrad_bnred_rows(N, H, W, K, C, code)
    p1 = torch.full((max(nrows, r1, 1), 2, C), float('nan'), device='cuda')
    call('kp_conv_dgrad_bnred', ptr(dy), ptr(wd), ptr(gx1), ptr(yred), *[ptr(t) for t in bn], slope, ptr(p1), N, H, W, K, C, code, st)
    assert r1 >= 1
    assert torch.isfinite(gx1.float()).all() and torch.equal(gx0, gx1)
    s0 = p0[:r0].double().sum(0).cpu()
    s1 = p1[:r1].double().sum(0).cpu()
    assert torch.isfinite(p1[:r1]).all()
    # scale of the sums: sum of |terms| ~ N*H*W * |gh| (x |xhat| for the second sum); two orders of the same fp32 additions:
    # 2e-5 of that (the bound the BatchNorm statistics are held to elsewhere); one misplaced pixel is ~1 / (N H W) >= 1e-4 here
    ref = yred.float()
    scale = float(gx0.float().abs().sum() / C) * max(1.0, float(ref.abs().max()))
    assert (s0 - s1).abs().max() <= 2e-5 * scale + 1e-6, ((s0 - s1).abs().max(), scale)
    # ... and an independent fp64 restatement of the two sums from the header's definition (include/kp_hip.h: gh = fold(gx) * act'(z),
    # sums of gh and gh * xhat over layer L - 1's conv output with ITS BatchNorm; fold = adjoint of ReplicationPad2d(1), vgg.py:30)
    gxd = gx1.double().cpu()
    ridx = (torch.arange(H + 2) - 1).clamp(0, H - 1)
    cidx = (torch.arange(W + 2) - 1).clamp(0, W - 1)
    gh = torch.zeros(N, H, W + 2, C, dtype=torch.float64).index_add_(1, ridx, gxd)
    gh = torch.zeros(N, H, W, C, dtype=torch.float64).index_add_(2, cidx, gh)
    yd = yred.double().cpu()
    if bn_on:
        mean, invstd, gamma, beta = [t.double().cpu() for t in bn]
        xhat = (yd - mean) * invstd
        z = xhat * gamma + beta
    else:
        xhat = z = yd
    gh = gh * torch.where(z > 0, 1.0, float(slope))
    want = torch.stack((gh.sum(dim=(0, 1, 2)), (gh * xhat).sum(dim=(0, 1, 2))))
    assert (s1 - want).abs().max() <= 2e-5 * scale + 1e-6, ((s1 - want).abs().max(), scale)


def test_unsupported_shapes_launch_nothing():
    """Shapes without a fused kernel (model F's 64..512-channel layers, fp32 storage) answer 0 / KP_E_UNSUPPORTED."""
    from keypoints_amd import _lib
    L = _lib.lib()
    assert L.kp_conv_bnin_supported(2, 16, 16, 64, 128, _lib.KP_BF16) == 0
    assert L.kp_conv_bnin_supported(2, 84, 84, 16, 32, _lib.KP_F32) == 0
    assert L.kp_conv_bnin_supported(2, 84, 84, 16, 32, _lib.KP_BF16) == 1
    assert L.kp_head_bnin_supported(2, 16, 16, 512, 64, _lib.KP_BF16) == 0      # model F's wide encoder head
    assert L.kp_head_bnin_supported(2, 84, 84, 32, 16, _lib.KP_F32) == 0
    assert L.kp_head_bnin_supported(2, 128, 128, 64, 3, _lib.KP_BF16) == 1       # model F's decoder head
    assert L.kp_head_bnin_supported(256, 84, 84, 32, 16, _lib.KP_F16) == 1       # Pong-class encoder head
    y = torch.zeros(8, device='cuda')
    with pytest.raises(_lib.KpError):
        _lib.call('kp_conv_fwd_bnin', _lib.ptr(y), None, None, None, None, 0.0, _lib.ptr(y), None, _lib.ptr(y), None,
                  2, 16, 16, 64, 128, _lib.KP_BF16, None)


@pytest.mark.parametrize('typ,cin,z,K,size', [('VGG_PONG_LAYERNECK', 1, 16, 4, 84), ('VGG_PACMAN', 3, 16, 4, 37),
                                              ('VGG_PONG_TRIVIAL', 1, 16, 3, 32)])
@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_train_steps_are_bitwise_the_same_with_and_without_the_fusion(typ, cin, z, K, size, precision):
    """Two Adam steps of a Pong-class Transporter: outputs, loss, every parameter and BatchNorm buffer bit-identical with the
    BatchNorm-apply pass fused into the consumers (default) and run as its own launch (runtime.set_fuse_bn_input(False))."""
    import keypoints_amd as ka
    from keypoints_amd import runtime
    from keypoints_amd.models import transporter
    g = torch.Generator().manual_seed(5)
    xs = (torch.rand(6, cin, size, size, generator=g) * 2 - 1).cuda()
    xt = (torch.rand(6, cin, size, size, generator=g) * 2 - 1).cuda()
    res = []
    try:
        for fuse in (False, True):
            runtime.set_fuse_bn_input(fuse)
            torch.manual_seed(9)
            net = transporter.make(typ, cin, z, K).cuda().set_precision(precision)
            opt = ka.FusedAdam(net.parameters(), lr=1e-3).bind(net)
            if precision == 'fp16':
                opt.scaler = ka.amp.LossScaler('cuda')
            for _ in range(2):
                opt.zero_grad()
                out = net(xs, xt)
                loss = ka.l2_reconstruction_loss(out[0], xt)
                with ka.amp.scale_loss(loss, opt) as sl:
                    sl.backward()
                opt.step()
            torch.cuda.synchronize()
            res.append((out[0].detach().clone(), loss.detach().clone(), {k: v.detach().clone() for k, v in net.state_dict().items()}))
    finally:
        runtime.set_fuse_bn_input(True)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_train_step_with_the_reduce_pass_in_the_data_gradient(precision):
    """runtime.set_fuse_bn_reduce (on by default) against the separate pass: a train step of the Pong transporter with the first pass of the
    BatchNorm backward in the data gradients' epilogues -- same loss bit for bit (the forward is untouched), parameter gradients to
    fp32-summation accuracy of the sums that feed them."""
    import keypoints_amd as ka
    from keypoints_amd import runtime
    from keypoints_amd.models import transporter
    g = torch.Generator().manual_seed(11)
    xs = (torch.rand(6, 1, 84, 84, generator=g) * 2 - 1).cuda()
    xt = (torch.rand(6, 1, 84, 84, generator=g) * 2 - 1).cuda()
    res = []
    try:
        for fuse in (False, True):
            runtime.set_fuse_bn_reduce(fuse)
            torch.manual_seed(9)
            net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 4).cuda().set_precision(precision)
            out = net(xs, xt)
            loss = ka.l2_reconstruction_loss(out[0], xt)
            loss.backward()
            torch.cuda.synchronize()
            res.append((loss.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}))
    finally:
        runtime.set_fuse_bn_reduce(True)
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys() and len(res[0][1]) > 10
    worst = 0.0
    for k in res[0][1]:
        a, b = res[0][1][k].float(), res[1][1][k].float()
        assert torch.isfinite(b).all()
        worst = max(worst, float((a - b).norm() / (a.norm() + 1e-12)))
    # the 16-bit rounding of dy downstream of slightly different sums: a few units of the storage type's epsilon, relative L2
    assert worst <= (2e-2 if precision == 'bf16' else 3e-3), worst


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_backward_falls_back_when_the_library_declines_the_fused_form(precision, monkeypatch):
    """The forward skips an activation pass where the shape queries offer the fused consumers; a backward kernel may still answer
    KP_E_UNSUPPORTED (unaligned gradient tensors, a measurement knob that changed in between).  unit_engine then writes the
    activation out (kp_bn_act_fwd of the saved conv output) and runs the plain kernels: same gradients, bit for bit."""
    import keypoints_amd as ka
    from keypoints_amd import unit_engine
    from keypoints_amd.models import transporter
    g = torch.Generator().manual_seed(13)
    xs = (torch.rand(5, 1, 84, 84, generator=g) * 2 - 1).cuda()
    xt = (torch.rand(5, 1, 84, 84, generator=g) * 2 - 1).cuda()
    real = unit_engine.try_call
    declined = []

    def declining(name, *args):
        declined.append(name)
        return False      # as if the library had returned KP_E_UNSUPPORTED (nothing launched)

    res = []
    for decline in (False, True):
        monkeypatch.setattr(unit_engine, 'try_call', declining if decline else real)
        torch.manual_seed(9)
        net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 4).cuda().set_precision(precision)
        loss = ka.l2_reconstruction_loss(net(xs, xt)[0], xt)
        loss.backward()
        torch.cuda.synchronize()
        res.append((loss.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}))
    assert {'kp_head_bwd_bnin', 'kp_conv_wgrad_bnin'} <= set(declined)
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


@pytest.mark.parametrize('N,H,W,K,C', [(128, 128, 128, 128, 64), (128, 64, 64, 256, 256), (128, 32, 32, 512, 512), (128, 16, 16, 512, 512),
                                       (128, 16, 16, 512, 64), (128, 64, 64, 256, 128), (128, 128, 128, 64, 128)])
def test_dgrad_bnred_at_the_bench_shapes(N, H, W, K, C):
    """Model F's data gradients at the bench batch (128 images), whole device: the fused sums against the separate pass."""
    from keypoints_amd._lib import call
    call('kp_config', 9, 2)
    try:
        _check_dgrad_bnred(N, H, W, K, C, 'fp16', True, 0.01)
    finally:
        call('kp_config', 9, 1)
