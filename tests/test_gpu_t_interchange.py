"""16-bit NHWC ("T") interchange between the feature encoder's 1x1 head, the fused keypoint bottleneck and the decoder
(include/kp_hip.h: kp_head_fwd_t / kp_head_bwd_t / kp_keypoint_bottleneck_fwd_t / _bwd_t / kp_nhwc_to_nchw), against the fp32
NCHW entry points that mirror the reference's Unit interface (knn.py:127-130, transporter.py:34-64) and are themselves pinned to
the oracle (test_gpu_layers.py, test_gpu_bottleneck*.py).  Same arithmetic, another layout: wherever the two forms see the same
values they must agree bit for bit (the T outputs being the fp32 results rounded once to the storage type); sums whose order
differs (bias gradient, the heat-map gradient's channel sum) agree to fp32 summation accuracy."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {'bf16': torch.bfloat16, 'fp16': torch.float16}


def _code(dt):
    from keypoints_amd import _lib
    return _lib.KP_BF16 if dt == 'bf16' else _lib.KP_F16


def _bn(C, g, on=True):
    if not on:
        return (None,) * 4
    return ((torch.randn(C, generator=g) * 0.3).cuda(), (torch.rand(C, generator=g) + 0.5).cuda(),
            (torch.randn(C, generator=g) * 0.5 + 1).cuda(), (torch.randn(C, generator=g) * 0.3).cuda())


def _nchw(t):   # T [N][H][W][K] -> fp32 NCHW (exact)
    return t.float().permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize('N,H,W', [(4, 84, 84), (2, 16, 24), (1, 8, 4), (3, 32, 32)])
@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('mode', ['plain', 'bnin', 'bnin_nobn'])
def test_head_t_equals_the_fp32_plane_form(N, H, W, dt, mode):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    C, K = 32, 16
    tdt, code = DT[dt], _code(dt)
    L = _lib.lib()
    assert L.kp_head_t_supported(N, H, W, C, K, code) == 1
    assert L.kp_head_t_supported(N, H, W, C, K, _lib.KP_F32) == 0 and L.kp_head_t_supported(N, H, W, 64, K, code) == 0
    g = torch.Generator().manual_seed(N * 100 + H + (dt == 'fp16'))
    x = torch.randn(N, H, W, C, generator=g).to(tdt).cuda()
    bn = _bn(C, g, mode == 'bnin')
    slope_in = 0.01
    w = (torch.randn(K, C, generator=g) * 0.3).cuda()
    bias = (torch.randn(K, generator=g) * 0.2).cuda()
    slope = 0.01
    st = stream_ptr()
    bnin = mode != 'plain'
    bnp = [ptr(t) for t in bn]
    # forward
    out32 = torch.full((N, K, H, W), float('nan'), device='cuda')
    if bnin:
        call('kp_head_fwd_bnin', ptr(x), *bnp, slope_in, ptr(w), ptr(bias), slope, ptr(out32), N, H, W, C, K, code, st)
    else:
        call('kp_head_fwd', ptr(x), ptr(w), ptr(bias), slope, ptr(out32), N, H, W, C, K, code, st)
    out_t = torch.full((N, H, W, K), float('nan'), device='cuda', dtype=tdt)
    call('kp_head_fwd_t', ptr(x), int(bnin), *bnp, slope_in, ptr(w), ptr(bias), slope, ptr(out_t), N, H, W, C, K, code, st)
    assert torch.isfinite(out_t.float()).all()
    assert torch.equal(out_t, out32.permute(0, 2, 3, 1).to(tdt))
    # backward: the fp32 form on the very values the T form reads
    gout_t = (torch.randn(N, H, W, K, generator=g) * 0.1).to(tdt).cuda()
    ws_n = L.kp_head_bwd_ws_floats(N, H, W, C, K, code)
    res = []
    for t_form in (False, True):
        dx = torch.full((N, H, W, C), float('nan'), device='cuda', dtype=tdt)
        dw = torch.zeros(K, C, device='cuda')
        db = torch.zeros(K, device='cuda')
        ws = torch.full((ws_n,), float('nan'), device='cuda')
        if t_form:
            call('kp_head_bwd_t', ptr(gout_t), ptr(out_t), ptr(x), int(bnin), *bnp, slope_in, ptr(w), slope, ptr(dx), ptr(dw), ptr(db),
                 ptr(ws), N, H, W, C, K, code, st)
        else:
            g32, o32 = _nchw(gout_t), _nchw(out_t)
            if bnin:
                call('kp_head_bwd_bnin', ptr(g32), ptr(o32), ptr(x), *bnp, slope_in, ptr(w), slope, ptr(dx), ptr(dw), ptr(db), ptr(ws),
                     N, H, W, C, K, code, st)
            else:
                call('kp_head_bwd', ptr(g32), ptr(o32), ptr(x), ptr(w), slope, ptr(dx), ptr(dw), ptr(db), ptr(ws), N, H, W, C, K, code, st)
        res.append((dx.float().cpu(), dw.cpu(), db.cpu()))
    assert all(torch.isfinite(t).all() for t in res[1])
    assert torch.equal(res[0][0], res[1][0])      # input gradient: same operands into the same MFMAs
    assert torch.equal(res[0][1], res[1][1])      # weight gradient: likewise, same partial / reduction order
    assert (res[0][2] - res[1][2]).abs().max() <= 2e-5 * max(1.0, float(res[0][2].abs().max()) * 50)   # bias gradient: another summation order
    # atomics mode (ws == NULL) of the T form: same sums up to the order
    dw2 = torch.zeros(K, C, device='cuda')
    db2 = torch.zeros(K, device='cuda')
    call('kp_head_bwd_t', ptr(gout_t), ptr(out_t), ptr(x), int(bnin), *bnp, slope_in, ptr(w), slope, None, ptr(dw2), ptr(db2), None,
         N, H, W, C, K, code, st)
    assert (dw2.cpu() - res[1][1]).abs().max() <= 1e-4 * max(1.0, float(res[1][1].abs().max()))
    assert (db2.cpu() - res[1][2]).abs().max() <= 1e-4 * max(1.0, float(res[1][2].abs().max()))


def _fold(ga):
    """adjoint of the replicate pad in the summation order of kp_nhwc_grad_to_nchw: rows outer, columns inner, from 0"""
    B, Hp, Wp, C = ga.shape
    H, W = Hp - 2, Wp - 2
    g = ga.float()
    out = torch.zeros(B, H, W, C)
    rows = lambda i: range(0 if i == 0 else i + 1, (H + 1 if i == H - 1 else i + 1) + 1)
    cols = lambda j: range(0 if j == 0 else j + 1, (W + 1 if j == W - 1 else j + 1) + 1)
    for i in range(H):
        for j in range(W):
            a = torch.zeros(B, C)
            for pr in rows(i):
                for pc in cols(j):
                    a = a + g[:, pr, pc]
            out[:, i, j] = a
    return out


@pytest.mark.parametrize('B,C,K,H,W', [(3, 16, 4, 84, 84), (2, 8, 3, 9, 13), (2, 32, 10, 32, 32), (1, 16, 1, 5, 6), (2, 16, 5, 16, 16)])
@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
def test_bottleneck_t_equals_the_fp32_plane_form(B, C, K, H, W, dt):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    tdt, code = DT[dt], _code(dt)
    L = _lib.lib()
    assert L.kp_keypoint_bottleneck_t_supported(B, C, K, H, W, code) == 1
    assert L.kp_keypoint_bottleneck_t_supported(B, 128, K, H, W, code) == 0 and L.kp_keypoint_bottleneck_t_supported(B, C, K, H, W, _lib.KP_F32) == 0
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + C + K)
    heat_s = (torch.randn(B, K, H, W, generator=g) * 3).cuda()
    heat_t = (torch.randn(B, K, H, W, generator=g) * 3).cuda()
    phi_s = torch.randn(B, H, W, C, generator=g).to(tdt).cuda()
    phi_t = torch.randn(B, H, W, C, generator=g).to(tdt).cuda()
    st = stream_ptr()
    sigma, eps = 0.1, 1e-6
    f32 = dict(device='cuda', dtype=torch.float32)

    def outs():
        return dict(k=torch.full((B, K, 2), float('nan'), **f32), p=torch.full((B, K, H), float('nan'), **f32),
                    q=torch.full((B, K, W), float('nan'), **f32), m=torch.full((B, K, H, W), float('nan'), **f32),
                    ms=torch.full((B, 1, H, W), float('nan'), **f32), mt=torch.full((B, 1, H, W), float('nan'), **f32),
                    arg=torch.full((B, H, W), 255, device='cuda', dtype=torch.uint8))
    r, t = outs(), outs()
    phi = torch.full((B, C, H, W), float('nan'), **f32)
    ps32, pt32 = _nchw(phi_s), _nchw(phi_t)
    call('kp_keypoint_bottleneck_fwd', ptr(heat_s), ptr(heat_t), ptr(ps32), ptr(pt32), ptr(r['k']), ptr(r['p']), ptr(r['q']), ptr(r['m']),
         ptr(phi), ptr(r['ms']), ptr(r['mt']), ptr(r['arg']), B, C, K, H, W, sigma, eps, st)
    a = torch.full((B, H + 2, W + 2, C), float('nan'), device='cuda', dtype=tdt)
    call('kp_keypoint_bottleneck_fwd_t', ptr(heat_s), ptr(heat_t), ptr(phi_s), ptr(phi_t), ptr(t['k']), ptr(t['p']), ptr(t['q']), ptr(t['m']),
         ptr(a), ptr(t['ms']), ptr(t['mt']), ptr(t['arg']), B, C, K, H, W, sigma, eps, code, st)
    for key in r:
        assert torch.equal(r[key], t[key]), key
    assert torch.isfinite(a.float()).all()
    want = phi.permute(0, 2, 3, 1).to(tdt)                                     # the fp32 blend rounded once
    pad = torch.nn.functional.pad(want.float().permute(0, 3, 1, 2), (1, 1, 1, 1), mode='replicate').permute(0, 2, 3, 1).to(tdt)
    assert torch.equal(a, pad)
    # kp_nhwc_to_nchw: the interior back as fp32 planes
    back = torch.full((B, C, H, W), float('nan'), **f32)
    call('kp_nhwc_to_nchw', ptr(a), ptr(back), B, C, H, W, 1, code, st)
    assert torch.equal(back, want.float().permute(0, 3, 1, 2))

    # backward
    ga = (torch.randn(B, H + 2, W + 2, C, generator=g) * 0.1).to(tdt).cuda()
    gk = (torch.randn(B, K, 2, generator=g) * 0.1).cuda()
    gm = (torch.randn(B, K, H, W, generator=g) * 0.01).cuda()
    for with_extra in (False, True):
        g_k, g_m = (ptr(gk), ptr(gm)) if with_extra else (None, None)
        d_phi_t = torch.full((B, H, W, C), float('nan'), device='cuda', dtype=tdt)
        d_heat_t = torch.full((B, K, H, W), float('nan'), **f32)
        call('kp_keypoint_bottleneck_bwd_t', ptr(ga), g_k, g_m, ptr(phi_s), ptr(phi_t), ptr(t['ms']), ptr(t['mt']), ptr(t['arg']),
             ptr(t['k']), ptr(t['p']), ptr(t['q']), ptr(d_phi_t), ptr(d_heat_t), B, C, K, H, W, sigma, eps, code, st)
        gphi = torch.full((B, C, H, W), float('nan'), **f32)
        call('kp_nhwc_grad_to_nchw', ptr(ga), ptr(gphi), B, C, C, H, W, 1, code, st)
        d_phi = torch.full((B, C, H, W), float('nan'), **f32)
        d_heat = torch.full((B, K, H, W), float('nan'), **f32)
        call('kp_keypoint_bottleneck_bwd', ptr(gphi), g_k, g_m, ptr(ps32), ptr(pt32), ptr(r['ms']), ptr(r['mt']), ptr(r['arg']),
             ptr(r['k']), ptr(r['p']), ptr(r['q']), ptr(d_phi), ptr(d_heat), B, C, K, H, W, sigma, eps, st)
        assert torch.isfinite(d_phi_t.float()).all() and torch.isfinite(d_heat_t).all()
        assert torch.equal(d_phi_t, d_phi.permute(0, 2, 3, 1).to(tdt))
        scale = max(1e-6, float(d_heat.abs().max()))
        assert (d_heat_t - d_heat).abs().max() <= 2e-5 * scale * max(1, C // 4), float((d_heat_t - d_heat).abs().max() / scale)
    if H * W <= 200:   # the fold itself against an independent restatement
        gphi = torch.empty(B, C, H, W, **f32)
        call('kp_nhwc_grad_to_nchw', ptr(ga), ptr(gphi), B, C, C, H, W, 1, code, st)
        assert torch.equal(gphi.cpu(), _fold(ga.cpu()).permute(0, 3, 1, 2))


def _pong_net(dt, seed=3):
    from keypoints_amd.models import transporter
    torch.manual_seed(seed)
    net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 4, combine_mode='max').cuda()
    net.set_precision(dt)
    return net


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
def test_transporter_takes_the_t_route_and_agrees_with_the_fp32_route(dt):
    """A train step of the Pong-class Transporter (configs/transporter_pong_grey.yaml shapes) with the T interchange on and off:
    the route is taken (no layout kernel output: phi comes back T-rounded), the two steps agree to the storage type's rounding
    -- the T route rounds the head's output once before the blend, the fp32 route rounds the blend once."""
    import keypoints_amd as ka
    from keypoints_amd.optim import l2_reconstruction_loss
    B, H, W = 8, 84, 84
    g = torch.Generator().manual_seed(11)
    xs = torch.rand(B, 1, H, W, generator=g).cuda()
    xt = torch.rand(B, 1, H, W, generator=g).cuda()
    res = []
    for on in (False, True):
        ka.set_t_interchange(on)
        try:
            net = _pong_net(dt)
            assert (net._t_route(xt) is not None) == on
            out = net(xs, xt)
            loss = l2_reconstruction_loss(out[0], xt)
            loss.backward()
            torch.cuda.synchronize()
            grads = torch.cat([p.grad.flatten() for p in net.parameters()]).cpu()
            flat = [o for m in out[3:] for o in (m if isinstance(m, (tuple, list)) else (m,))]
            res.append((out[0].detach().cpu(), out[1].detach().cpu(), out[2].detach().cpu(), float(loss), grads,
                        [o.detach().cpu() for o in flat]))
        finally:
            ka.set_t_interchange(True)
    (x0, phi0, k0, l0, g0, rest0), (x1, phi1, k1, l1, g1, rest1) = res
    assert phi1.shape == phi0.shape and phi1.dtype == torch.float32
    assert torch.equal(phi1, phi1.to(DT[dt]).float())          # the returned blend is the decoder's 16-bit input
    assert torch.equal(k0, k1)                                 # keypoints do not depend on the feature layout
    for a, b in zip(rest0, rest1):
        assert torch.equal(a, b)                               # m_t, (p, q) marginals, heat-map, masks
    ulp = 2.0 ** (-7 if dt == 'bf16' else -10)
    assert (phi1 - phi0).abs().max() <= 2 * ulp * max(1.0, float(phi0.abs().max()))
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-12))
    assert rel(x1, x0) <= 6 * ulp and abs(l1 - l0) <= 4 * ulp * abs(l0)
    assert rel(g1, g0) <= 20 * ulp


def test_t_route_is_declined_where_the_kernels_do_not_exist():
    from keypoints_amd.models import transporter
    torch.manual_seed(0)
    x = torch.rand(2, 1, 84, 84).cuda()
    net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 4, combine_mode='max').cuda()
    assert net._t_route(x) is None                                  # fp32 storage
    net.set_precision('bf16')
    assert net._t_route(x) is not None
    assert net._t_route(torch.rand(2, 1, 21, 21).cuda()) is None   # 882 pixels: not a multiple of 32
    net.combine_method = 'loop'
    assert net._t_route(x) is None
    net2 = transporter.make('VGG_PONG_LAYERNECK', 1, 12, 4, combine_mode='max').cuda().set_precision('bf16')
    assert net2._t_route(x) is None                                 # 12 feature channels


def test_gradient_through_the_returned_blend_reaches_the_encoder():
    """forward() returns the blend as an fp32 NCHW tensor (transporter.py:64); a loss on it must flow back into the T route."""
    net = _pong_net('bf16')
    g = torch.Generator().manual_seed(5)
    xs = torch.rand(4, 1, 84, 84, generator=g).cuda()
    xt = torch.rand(4, 1, 84, 84, generator=g).cuda()
    out = net(xs, xt)
    out[1].square().mean().backward()
    torch.cuda.synchronize()
    gsum = sum(float(p.grad.abs().sum()) for p in net.feature.parameters())
    assert gsum > 0 and all(torch.isfinite(p.grad).all() for p in net.feature.parameters())
    assert all(p.grad is None or float(p.grad.abs().sum()) == 0 for p in net.decoder.parameters())


def test_training_through_the_t_route_follows_the_fp32_plane_route():
    """60 Adam steps of the Pong-class Transporter on structured pairs (tests/structured.py), bf16, once with the 16-bit NHWC
    interchange + pair calls (the default) and once with the reference's fp32 NCHW tensors between the Units and two calls per
    encoder: the extra rounding of the feature maps must not show in the training curve -- both fall > 10x and every step's loss
    stays within 12 % of the other route's.  (Adam's first, sign-like steps amplify last-bit differences: the same comparison moved
    from 3.2 % to 6.0 % of the worst step when nothing but the summation order of the BatchNorm statistics rows changed in BOTH
    routes -- one row per band-kernel workgroup instead of four --, so that is the resolution of this test; a route that computed
    something else leaves the curve by whole multiples.)"""
    import keypoints_amd as ka
    from keypoints_amd.models import transporter
    from structured import structured_pairs
    curves = []
    for on in (True, False):
        ka.set_t_interchange(on); ka.set_pair_calls(on)
        try:
            torch.manual_seed(3)
            net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 4, combine_mode='max').cuda().set_precision('bf16')
            assert (net._t_route(torch.zeros(32, 1, 84, 84, device='cuda')) is not None) == on
            opt = ka.FusedAdam(net.parameters(), lr=1e-3).bind(net)
            g = torch.Generator().manual_seed(17)
            losses = []
            for it in range(60):
                xs, xt = structured_pairs(32, 84, g, channels=1, device='cuda')
                opt.zero_grad()
                out = net(xs, xt)
                loss = ka.l2_reconstruction_loss(out[0], xt)
                loss.backward()
                opt.step()
                losses.append(float(loss))
            curves.append(losses)
        finally:
            ka.set_t_interchange(True); ka.set_pair_calls(True)
    a, b = curves
    assert a[-1] < 0.1 * a[0] and b[-1] < 0.1 * b[0], (a[0], a[-1], b[0], b[-1])
    worst = max(abs(x - y) / y for x, y in zip(a, b))
    assert worst <= 0.12, worst


def test_returned_blend_is_made_on_first_use_and_equals_the_eager_copy():
    """forward()'s second output on the T route is a functional.LazyNCHW: a torch.Tensor by every test a caller can make (isinstance,
    shape, dtype, device) whose values appear at first use -- bit-identical to the eager copy of set_lazy_outputs(False), usable after
    the step's backward (logging), in arithmetic with other tensors, and as the input of another Unit."""
    import keypoints_amd as ka
    from keypoints_amd.models import functional as MF
    from keypoints_amd.optim import l2_reconstruction_loss
    g = torch.Generator().manual_seed(23)
    xs = torch.rand(4, 1, 84, 84, generator=g).cuda()
    xt = torch.rand(4, 1, 84, 84, generator=g).cuda()
    vals = []
    for lazy in (True, False):
        ka.set_lazy_outputs(lazy)
        try:
            net = _pong_net('bf16')
            out = net(xs, xt)
            phi = out[1]
            assert isinstance(phi, torch.Tensor) and (type(phi) is MF.LazyNCHW) == lazy
            assert phi.shape == (4, 16, 84, 84) and phi.dtype == torch.float32 and phi.device == xs.device and phi.dim() == 4
            if lazy:
                assert phi._kp_val is None                      # nothing has been computed for it yet
            l2_reconstruction_loss(out[0], xt).backward()       # the train step never touches it
            if lazy:
                assert phi._kp_val is None
            vals.append((phi.detach().cpu(), (phi * 2 + 1).sum().item(), float(phi[1, 2, 3, 4]), net.decoder(phi).detach().cpu()))
            if lazy:
                assert phi._kp_val is not None and phi._kp_src is None
        finally:
            ka.set_lazy_outputs(True)
    (a0, s0, e0, d0), (a1, s1, e1, d1) = vals
    assert torch.equal(a0, a1) and s0 == s1 and e0 == e1 and torch.equal(d0, d1)


def test_lazy_blend_first_touched_under_no_grad_still_carries_its_graph_afterwards():
    """A first use under torch.no_grad() (image logging between forward and backward) must not pin a graph-less copy: a later
    differentiable use of the returned map -- a loss on phi, transporter.py:64's second output -- reaches the encoders, with the same
    gradients as when the map is used with autograd on from the start."""
    import keypoints_amd as ka
    from keypoints_amd.models import functional as MF
    g = torch.Generator().manual_seed(29)
    xs = torch.rand(4, 1, 84, 84, generator=g).cuda()
    xt = torch.rand(4, 1, 84, 84, generator=g).cuda()
    grads = []
    for touch_first in (True, False):
        net = _pong_net('bf16')
        out = net(xs, xt)
        phi = out[1]
        assert type(phi) is MF.LazyNCHW
        if touch_first:
            with torch.no_grad():
                logged = phi.clamp(0, 1).sum().item()           # no graph: serves the log
            assert phi._kp_detached and phi._kp_src is not None
            with torch.no_grad():
                assert phi.clamp(0, 1).sum().item() == logged    # (the graph-less copy is reused under no_grad)
        loss = (phi * phi).mean()
        assert loss.requires_grad
        loss.backward()
        assert not phi._kp_detached and phi._kp_src is None
        gr = [p.grad.detach().clone() for p in net.feature.parameters() if p.grad is not None]
        assert gr and any(float(t.abs().max()) > 0 for t in gr)
        grads.append(gr)
    assert len(grads[0]) == len(grads[1]) and all(torch.equal(a, b) for a, b in zip(*grads))
