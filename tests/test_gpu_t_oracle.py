"""The 16-bit NHWC ("T") kernels of round 4 against the CPU oracle directly (test_gpu_t_interchange.py pins them to the fp32 plane
forms bit for bit; this file is the independent leg: the same entry points against fp64 arithmetic of oracle/kp_oracle.py on the
very 16-bit values the kernels read, with torch autograd through the oracle for the gradients).

Tolerances.  A T output is the fp32 result rounded ONCE to the storage type: |out - exact| <= ulp_T(exact) / 2 + fp32 noise, asserted
as one whole ulp (2^-8 relative for bf16, 2^-11 for fp16, + a small absolute term for values near zero).  fp32 outputs (keypoints,
marginals, maps, heat-map gradient) use SURVEY 8c's fp32 op-level bounds (atol 1e-6 / rtol 1e-5; the heat-map gradient is a sum of
H*W terms with 1/d up to 1e3: 2e-4 of its scale, as in test_gpu_bottleneck_fused.py)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {'bf16': torch.bfloat16, 'fp16': torch.float16}
ULP = {'bf16': 2.0 ** -8, 'fp16': 2.0 ** -11}


def _code(dt):
    from keypoints_amd import _lib
    return _lib.KP_BF16 if dt == 'bf16' else _lib.KP_F16


def _one_ulp(got_t, exact, dt, floor):
    """got_t (storage type) is `exact` (fp64) rounded once: within one ulp of the type (relative), `floor` absolute near zero."""
    err = (got_t.double().cpu() - exact).abs()
    bound = ULP[dt] * exact.abs() + floor
    assert bool((err <= bound).all()), float((err - bound).max())


@pytest.mark.parametrize('B,C,K,H,W', [(2, 16, 4, 84, 84), (2, 8, 3, 9, 13), (2, 32, 10, 32, 32), (3, 16, 5, 16, 16)])
@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('extra', [False, True])
def test_bottleneck_t_against_the_oracle(B, C, K, H, W, dt, extra):
    """kp_keypoint_bottleneck_fwd_t / _bwd_t == functional.py:11-63 + transporter.py:57-60 ('max') in fp64 on the same 16-bit feature
    maps; the decoder's padded input is the blend with ReplicationPad2d(1) (vgg.py:30), its gradient arrives over the padded domain."""
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    from oracle import kp_oracle as O
    tdt, code = DT[dt], _code(dt)
    g = torch.Generator().manual_seed(B * 977 + H * 13 + C + K + (dt == 'fp16'))
    heat_s = torch.randn(B, K, H, W, generator=g) * 3
    heat_t = torch.randn(B, K, H, W, generator=g) * 3
    phi_s = torch.randn(B, H, W, C, generator=g).to(tdt)
    phi_t = torch.randn(B, H, W, C, generator=g).to(tdt)
    st = stream_ptr()
    sigma, eps = 0.1, 1e-6
    f32 = dict(device='cuda', dtype=torch.float32)
    k = torch.full((B, K, 2), float('nan'), **f32)
    p = torch.full((B, K, H), float('nan'), **f32)
    q = torch.full((B, K, W), float('nan'), **f32)
    m = torch.full((B, K, H, W), float('nan'), **f32)
    ms = torch.full((B, 1, H, W), float('nan'), **f32)
    mt = torch.full((B, 1, H, W), float('nan'), **f32)
    arg = torch.full((B, H, W), 255, device='cuda', dtype=torch.uint8)
    a = torch.full((B, H + 2, W + 2, C), float('nan'), device='cuda', dtype=tdt)
    hs_d, ht_d, ps_d, pt_d = heat_s.cuda(), heat_t.cuda(), phi_s.cuda(), phi_t.cuda()
    call('kp_keypoint_bottleneck_fwd_t', ptr(hs_d), ptr(ht_d), ptr(ps_d), ptr(pt_d), ptr(k), ptr(p), ptr(q), ptr(m),
         ptr(a), ptr(ms), ptr(mt), ptr(arg), B, C, K, H, W, sigma, eps, code, st)

    # oracle, fp64, with autograd
    oh = heat_t.double().requires_grad_(True)
    op = phi_t.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    with torch.no_grad():
        oks, _ = O.spatial_logsoftmax(heat_s.double())
        oms = O.gaussian_like(oks, H, W, sigma, eps)
    okt, (opr, oqr) = O.spatial_logsoftmax(oh)
    omt = O.gaussian_like(okt, H, W, sigma, eps)
    ophi, omask_s, omask_t = O.transport(phi_s.double().permute(0, 3, 1, 2), op, oms, omt, 'max')
    opad = F.pad(ophi, (1, 1, 1, 1), mode='replicate')

    def close32(got, want, atol=1e-6, rtol=1e-5):
        err = (got.double().cpu() - want.detach()).abs()
        assert bool((err <= atol + rtol * want.detach().abs()).all()), float(err.max())
    close32(k, okt)
    close32(p, opr.reshape(B, K, H))
    close32(q, oqr.reshape(B, K, W))
    close32(m, omt, atol=2e-5)   # exp(-50 d): d carries the keypoint's 1e-6
    close32(ms, omask_s.reshape(B, 1, H, W), atol=2e-5)
    close32(mt, omask_t.reshape(B, 1, H, W), atol=2e-5)
    _one_ulp(a, opad.detach().permute(0, 2, 3, 1), dt, floor=1e-4)
    # the arg-max plane names a keypoint whose map attains the maximum at that pixel
    picked = torch.gather(omt.detach(), 1, arg.cpu().long().unsqueeze(1)).squeeze(1)
    assert bool((picked >= omt.detach().max(dim=1).values - 2e-5).all())

    # backward: a gradient over the padded domain (+ direct gradients on the keypoints / maps)
    ga = (torch.randn(B, H + 2, W + 2, C, generator=g) * 0.1).to(tdt)
    gk = torch.randn(B, K, 2, generator=g) * 0.1
    gm = torch.randn(B, K, H, W, generator=g) * 0.01
    ga_d, gk_d, gm_d = ga.cuda(), gk.cuda(), gm.cuda()
    d_phi_t = torch.full((B, H, W, C), float('nan'), device='cuda', dtype=tdt)
    d_heat_t = torch.full((B, K, H, W), float('nan'), **f32)
    call('kp_keypoint_bottleneck_bwd_t', ptr(ga_d), ptr(gk_d) if extra else None, ptr(gm_d) if extra else None, ptr(ps_d), ptr(pt_d),
         ptr(ms), ptr(mt), ptr(arg), ptr(k), ptr(p), ptr(q), ptr(d_phi_t), ptr(d_heat_t), B, C, K, H, W, sigma, eps, code, st)
    oloss = (opad * ga.double().permute(0, 3, 1, 2)).sum()
    if extra:
        oloss = oloss + (okt * gk.double()).sum() + (omt * gm.double()).sum()
    oloss.backward()
    _one_ulp(d_phi_t, op.grad.permute(0, 2, 3, 1), dt, floor=1e-5)
    scale = float(oh.grad.abs().max())
    err = float((d_heat_t.double().cpu() - oh.grad).abs().max())
    assert err <= 2e-4 * scale + 1e-7, err / scale


@pytest.mark.parametrize('N,H,W', [(4, 84, 84), (2, 16, 24), (1, 8, 4)])
@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('bnin', [False, True])
def test_head_t_against_the_oracle(N, H, W, dt, bnin):
    """kp_head_fwd_t / kp_head_bwd_t == knn.py:123-125 (Conv2d 1x1 + LeakyReLU) in fp64; with the BatchNorm input, its operand is
    the previous layer's BatchNorm + LeakyReLU (vgg.py:33-38) of the 16-bit conv output, rounded to the storage type as the
    materialised activation would have been (kp_bn_act_fwd)."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    C, K = 32, 16
    tdt, code = DT[dt], _code(dt)
    L = _lib.lib()
    g = torch.Generator().manual_seed(N * 131 + H + 7 * (dt == 'fp16') + bnin)
    x = torch.randn(N, H, W, C, generator=g).to(tdt)
    mean, invstd = torch.randn(C, generator=g) * 0.3, torch.rand(C, generator=g) + 0.5
    gamma, beta = torch.randn(C, generator=g) * 0.5 + 1, torch.randn(C, generator=g) * 0.3
    slope_in, slope = 0.01, 0.01
    w = torch.randn(K, C, generator=g) * 0.3
    bias = torch.randn(K, generator=g) * 0.2
    st = stream_ptr()
    xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
    bn = [t.cuda() for t in (mean, invstd, gamma, beta)]
    bnp = [ptr(t) if bnin else None for t in bn]
    out_t = torch.full((N, H, W, K), float('nan'), device='cuda', dtype=tdt)
    call('kp_head_fwd_t', ptr(xd), int(bnin), *bnp, slope_in, ptr(wd), ptr(bd), slope, ptr(out_t), N, H, W, C, K, code, st)

    # oracle: the operand of the 1x1 convolution is a 16-bit activation
    if bnin:
        pre = (x.double() - mean.double()) * invstd.double() * gamma.double() + beta.double()
        act = F.leaky_relu(pre, slope_in).to(tdt).double()
    else:
        act = x.double()
    act = act.requires_grad_(True)
    ow = w.double().requires_grad_(True)
    ob = bias.double().requires_grad_(True)
    z = act @ ow.t() + ob
    oout = F.leaky_relu(z, slope)
    # the fp32 accumulation of 32 products + the rounding: one ulp of the type + 32 fp32 roundings of the largest term
    if not bnin:
        _one_ulp(out_t, oout.detach(), dt, floor=2e-5)
    else:
        # the kernel forms the activation in fp32 and rounds it to the storage type; where the fp32 and the fp64 value straddle a
        # rounding boundary the operand differs by one ulp of the type (a 2^-13 chance per element in fp16): hard bound with every
        # operand one ulp off, and at least 99 % of the outputs inside the plain one-ulp bound
        err = (out_t.double().cpu() - oout.detach()).abs()
        one = ULP[dt] * oout.detach().abs() + 2e-5
        hard = one + ULP[dt] * (act.detach().abs() @ w.double().abs().t())
        assert bool((err <= hard).all()), float((err - hard).max())
        assert float((err <= one).double().mean()) >= 0.99

    gout = (torch.randn(N, H, W, K, generator=g) * 0.1).to(tdt)
    gd = gout.cuda()
    ws = torch.full((L.kp_head_bwd_ws_floats(N, H, W, C, K, code),), float('nan'), device='cuda')
    dx = torch.full((N, H, W, C), float('nan'), device='cuda', dtype=tdt)
    dw = torch.zeros(K, C, device='cuda')
    db = torch.zeros(K, device='cuda')
    call('kp_head_bwd_t', ptr(gd), ptr(out_t), ptr(xd), int(bnin), *bnp, slope_in, ptr(wd), slope, ptr(dx), ptr(dw), ptr(db), ptr(ws),
         N, H, W, C, K, code, st)
    # the kernel takes the LeakyReLU branch from the sign of the stored output, i.e. of ITS fp32 sum: where |z| is inside the fp32
    # noise of a 32-term sum the branch is not defined by the oracle either -- those pixels are left out of the element-wise check
    (oout * gout.double()).sum().backward()
    sure = (z.detach().abs().min(dim=-1).values > 1e-4)
    assert float(sure.double().mean()) > 0.95
    # dx is the gradient at the convolution's operand (the BatchNorm backward is the producer's job, as for every 3x3 layer)
    _one_ulp(dx[sure.cuda()], act.grad[sure], dt, floor=3e-5)
    # weight / bias gradient: the MFMA operands are dz = gout * LeakyReLU'(z) and the activation, both in the storage type (what the
    # materialised chain multiplies, too): each of the N*H*W terms carries at most half an ulp of dz -> a hard bound from the oracle's
    # own |dz|, |a| (asserted at a whole ulp)
    dz = (gout.double() * torch.where(z.detach() > 0, 1.0, slope)).reshape(-1, K)
    a2 = act.detach().reshape(-1, C)
    bw = ULP[dt] * (dz.abs().t() @ a2.abs()) + 1e-5
    bb = ULP[dt] * dz.abs().sum(dim=0) + 1e-5
    assert bool(((dw.double().cpu() - ow.grad).abs() <= bw).all())
    assert bool(((db.double().cpu() - ob.grad).abs() <= bb).all())
    # ... and statistically far inside it (independent roundings): 2 % of the gradient's scale
    assert float((dw.double().cpu() - ow.grad).abs().max()) <= 2e-2 * float(ow.grad.abs().max())
    assert float((db.double().cpu() - ob.grad).abs().max()) <= 2e-2 * float(ob.grad.abs().max())


@pytest.mark.parametrize('C,K,H,W', [(16, 3, 5, 6), (16, 4, 16, 16), (8, 2, 3, 3)])
def test_bottleneck_t_small_maps_read_nothing_past_the_last_image(C, K, H, W):
    """Maps with H*W*(C/8) below the workgroup size (advisor, round 4): the forward's two-deep prefetch must clamp to the image.
    The feature maps are placed so that the LAST image ends exactly at the end of a dedicated 20 MiB device allocation (the caching
    allocator hands allocations above 10 MiB out at their own size, rounded to 2 MiB): a read past it leaves the allocation.  The
    results must equal those of the same call on tensors in the middle of an allocation."""
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    tdt, code, B = torch.bfloat16, _code('bf16'), 2
    g = torch.Generator().manual_seed(C + H)
    heat_s = (torch.randn(B, K, H, W, generator=g) * 3).cuda()
    heat_t = (torch.randn(B, K, H, W, generator=g) * 3).cuda()
    phi_s = torch.randn(B, H, W, C, generator=g).to(tdt)
    phi_t = torch.randn(B, H, W, C, generator=g).to(tdt)
    n = B * H * W * C
    outs = []
    for at_end in (False, True):
        if at_end:
            big_s = torch.empty(20 * 2 ** 20 // 2, device='cuda', dtype=tdt)
            big_t = torch.empty(20 * 2 ** 20 // 2, device='cuda', dtype=tdt)
            ps_d, pt_d = big_s[-n:].view(B, H, W, C), big_t[-n:].view(B, H, W, C)
            ps_d.copy_(phi_s); pt_d.copy_(phi_t)
            assert ps_d.data_ptr() + 2 * n == big_s.data_ptr() + 20 * 2 ** 20
        else:
            ps_d, pt_d = phi_s.cuda(), phi_t.cuda()
        f32 = dict(device='cuda', dtype=torch.float32)
        k = torch.empty(B, K, 2, **f32); p = torch.empty(B, K, H, **f32); q = torch.empty(B, K, W, **f32)
        m = torch.empty(B, K, H, W, **f32); ms = torch.empty(B, 1, H, W, **f32); mt = torch.empty(B, 1, H, W, **f32)
        arg = torch.empty(B, H, W, device='cuda', dtype=torch.uint8)
        a = torch.empty(B, H + 2, W + 2, C, device='cuda', dtype=tdt)
        call('kp_keypoint_bottleneck_fwd_t', ptr(heat_s), ptr(heat_t), ptr(ps_d), ptr(pt_d), ptr(k), ptr(p), ptr(q), ptr(m),
             ptr(a), ptr(ms), ptr(mt), ptr(arg), B, C, K, H, W, 0.1, 1e-6, code, stream_ptr())
        torch.cuda.synchronize()
        outs.append((k.clone(), m.clone(), a.float().clone(), arg.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
