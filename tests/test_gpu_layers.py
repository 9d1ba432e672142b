"""One Pad+Conv3x3+BN(train)+act[+pool|upsample] block and the 1x1 head through the HIP Unit engine,
against golden vectors produced by the reference's torch modules (tools/make_golden.py)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() / max(1e-12, np.abs(b).max())


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-5), ('bf16', 3e-2), ('fp16', 4e-3)])
@pytest.mark.parametrize('tag,cin,cout,tail,slope', [('in3', 3, 64, None, 0.01), ('c64', 32, 64, 'M', 0.01),
                                                      ('dec', 32, 16, 'U', 0.0), ('odd', 5, 7, None, 0.01)])
def test_block_against_golden(golden, tag, cin, cout, tail, slope, precision, tol):
    """Drive the engine op by op (same calls UnitFunction makes) on the golden block and compare the
    activation, every gradient and the running statistics."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import precision_dtype, stream_ptr
    z = golden('layers.npz')
    tdt, code = precision_dtype(precision)
    dev = 'cuda'
    x = torch.tensor(z[f'{tag}.x'], device=dev)
    N, C, H, W = x.shape
    K = cout
    wt = torch.tensor(z[f'{tag}.weight'], device=dev)
    bias = torch.tensor(z[f'{tag}.bias'], device=dev)
    gamma = torch.tensor(z[f'{tag}.gamma'], device=dev)
    beta = torch.tensor(z[f'{tag}.beta'], device=dev)
    st = stream_ptr()
    a = torch.empty(N, H + 2, W + 2, C, device=dev, dtype=tdt)
    call('kp_nchw_to_nhwc', ptr(x), ptr(a), N, C, C, H, W, 1, code, st)
    wf = torch.empty(K, 3, 3, C, device=dev, dtype=tdt)
    wd = torch.empty(C, 3, 3, K, device=dev, dtype=tdt)
    call('kp_pack_weights', ptr(wt), ptr(wf), ptr(wd), K, C, C, 3, code, st)
    y = torch.empty(N, H, W, K, device=dev, dtype=tdt)
    stats = torch.empty((N * H * W + 63) // 64, 2, K, device=dev, dtype=torch.float32)
    call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, code, st)
    mean = torch.empty(K, device=dev); invstd = torch.empty(K, device=dev)
    rm = torch.zeros(K, device=dev); rv = torch.ones(K, device=dev)
    nbt = torch.zeros((), device=dev, dtype=torch.long)
    ws = torch.empty(128 * K, device=dev, dtype=torch.float64)
    from keypoints_amd._lib import last_rows
    call('kp_bn_finalize', ptr(stats), last_rows(), ptr(ws), ptr(mean), ptr(invstd), ptr(rm), ptr(rv), ptr(nbt), 0.1, 1e-5, N * H * W, K, st)
    tcode = {None: 0, 'M': 1, 'U': 2}[tail]
    Ho, Wo = (H // 2, W // 2) if tail == 'M' else ((2 * H, 2 * W) if tail == 'U' else (H, W))
    out = torch.empty(N, Ho, Wo, K, device=dev, dtype=tdt)
    argmax = torch.empty(N, Ho, Wo, K, device=dev, dtype=torch.uint8) if tail == 'M' else None
    call('kp_bn_act_fwd', ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), slope, tcode, ptr(out), 0, ptr(argmax),
         N, H, W, K, code, st)
    got = out.float().permute(0, 3, 1, 2)
    assert relerr(got, z[f'{tag}.y']) < tol
    assert relerr(rm, z[f'{tag}.running_mean']) < max(tol, 1e-5) and relerr(rv, z[f'{tag}.running_var']) < max(tol, 1e-5)
    assert int(nbt.item()) == 1
    # padded output replicates its border
    outp = torch.empty(N, Ho + 2, Wo + 2, K, device=dev, dtype=tdt)
    call('kp_bn_act_fwd', ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), slope, tcode, ptr(outp), 1, None,
         N, H, W, K, code, st)
    ref = torch.nn.functional.pad(out.float().permute(0, 3, 1, 2), (1, 1, 1, 1), mode='replicate').permute(0, 2, 3, 1)
    assert torch.equal(outp.float(), ref)

    # backward with upstream gradient w (plain, no border)
    g = torch.tensor(z[f'{tag}.w'], device=dev).permute(0, 2, 3, 1).contiguous().to(tdt)
    if tail == 'U':
        g2 = torch.empty(N, H, W, K, device=dev, dtype=tdt)
        call('kp_upsample2_bwd', ptr(g), 0, ptr(g2), N, H, W, K, code, st)
        g, tcode_b = g2, 0
    else:
        tcode_b = tcode
    partials = torch.empty(N * H, 2, K, device=dev)
    call('kp_bn_act_bwd_reduce', ptr(g), 0, tcode_b, ptr(argmax), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
         slope, ptr(partials), N, H, W, K, code, st)
    dgamma = torch.zeros(K, device=dev); dbeta = torch.zeros(K, device=dev)
    sums = torch.empty(2 * K, device=dev)
    call('kp_bn_bwd_finalize', ptr(partials), ptr(ws), last_rows(), K, ptr(sums), ptr(dgamma), ptr(dbeta), st)
    dy = torch.empty(N, H + 4, W + 4, K, device=dev, dtype=tdt)
    call('kp_bn_act_bwd_apply', ptr(g), 0, tcode_b, ptr(argmax), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
         slope, ptr(sums), ptr(dy), 2, N, H, W, K, code, st)
    assert float(dy.float()[:, :2].abs().max()) == 0 and float(dy.float()[:, :, -2:].abs().max()) == 0
    parts = _lib.lib().kp_conv_wgrad_partials(N, H, W, C, K, 3, 2, code)
    scratch = torch.zeros(parts, K, 3, 3, C, device=dev)
    call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, W, C, K, 3, 2, code, st)
    dw = torch.zeros(K, C, 3, 3, device=dev)
    call('kp_wgrad_unpack', ptr(scratch), parts, ptr(dw), K, C, C, 3, st)
    gx = torch.empty(N, H + 2, W + 2, C, device=dev, dtype=tdt)
    call('kp_conv_fwd', ptr(dy), ptr(wd), None, ptr(gx), None, N, H + 4, W + 4, K, C, 3, code, st)
    dx = torch.empty(N, C, H, W, device=dev)
    call('kp_nhwc_grad_to_nchw', ptr(gx), ptr(dx), N, C, C, H, W, 1, code, st)
    # gradient sums of this 512-pixel block: in the 16-bit modes one pre-activation that crosses LeakyReLU's kink (or one
    # max-pool window that resolves differently) under the storage rounding of y moves a per-channel sum by several
    # per cent, whatever the mantissa width -- the bound there is for such single flips, not for rounding
    gt = tol * 5 if precision == 'fp32' else 0.15
    assert relerr(dgamma, z[f'{tag}.dgamma']) < gt and relerr(dbeta, z[f'{tag}.dbeta']) < gt
    assert relerr(dw, z[f'{tag}.dweight']) < gt
    assert relerr(dx, z[f'{tag}.dx']) < gt
    # conv bias in front of train-mode BN: reference gradient is rounding noise around 0
    assert np.abs(z[f'{tag}.dbias']).max() < 1e-4 * np.abs(z[f'{tag}.dweight']).max() + 1e-5


@pytest.mark.parametrize('precision,tol', [('fp32', 1e-5), ('bf16', 2e-2), ('fp16', 2.5e-3)])
def test_head_against_golden(golden, precision, tol):
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import precision_dtype, stream_ptr
    z = golden('layers.npz')
    tdt, code = precision_dtype(precision)
    dev = 'cuda'
    x = torch.tensor(z['head.x'], device=dev)
    N, C, H, W = x.shape
    K = 5
    a = torch.empty(N, H, W, C, device=dev, dtype=tdt)
    st = stream_ptr()
    call('kp_nchw_to_nhwc', ptr(x), ptr(a), N, C, C, H, W, 0, code, st)
    wt = torch.tensor(z['head.weight'], device=dev); b = torch.tensor(z['head.bias'], device=dev)
    out = torch.empty(N, K, H, W, device=dev)
    call('kp_head_fwd', ptr(a), ptr(wt), ptr(b), 0.01, ptr(out), N, H, W, C, K, code, st)
    assert relerr(out, z['head.y']) < tol
    gout = torch.tensor(z['head.w'], device=dev)
    dxn = torch.empty(N, H, W, C, device=dev, dtype=tdt)
    dw = torch.zeros(K, C, device=dev); db = torch.zeros(K, device=dev)
    call('kp_head_bwd', ptr(gout), ptr(out), ptr(a), ptr(wt), 0.01, ptr(dxn), ptr(dw), ptr(db), None, N, H, W, C, K, code, st)
    if precision == 'fp32':
        ref_dx, ref_dw, ref_db = z['head.dx'], z['head.dweight'], z['head.dbias']
    else:
        # 16-bit storage of x moves pre-activations that sit within rounding of 0 across LeakyReLU's kink (the golden
        # data has one at 1e-6): the gradient there legitimately switches between slope and 1.  The backward is
        # therefore judged against torch autograd on the SAME rounded activation (kernel arithmetic, not the kink).
        af = a.float().permute(0, 3, 1, 2).requires_grad_(True)
        wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(af, wr, br), 0.01)
        assert relerr(out, yr.detach().cpu().numpy()) < 1e-5
        yr.backward(gout)
        ref_dx, ref_dw, ref_db = af.grad.cpu().numpy(), wr.grad.cpu().numpy(), br.grad.cpu().numpy()
    assert relerr(dxn.float().permute(0, 3, 1, 2), ref_dx) < tol
    assert relerr(dw.view(K, C, 1, 1), ref_dw) < tol and relerr(db, ref_db) < tol


@pytest.mark.parametrize('N,H,W,C,K', [(2, 16, 16, 64, 3), (1, 8, 8, 512, 10), (3, 5, 7, 16, 1), (2, 32, 32, 128, 16),
                                       # kp_head_mfma.hip: C % 128 == 0, K <= 64, H*W % 32 == 0
                                       (2, 16, 16, 512, 64), (1, 8, 16, 256, 33), (3, 16, 16, 128, 48), (2, 8, 8, 128, 3),
                                       (64, 16, 16, 512, 64),
                                       # the 32-channel MFMA heads (Pong-class encoder heads): groups of 32 pixels that straddle
                                       # images (H*W % 32 != 0, H*W % 8 == 0), K up to 64; and a ragged one that stays on the VALU kernels
                                       (8, 84, 84, 32, 16), (4, 6, 12, 32, 16), (16, 10, 4, 32, 33), (8, 84, 84, 32, 5), (3, 7, 9, 32, 16)])
def test_head_small_k_bf16(N, H, W, C, K):
    """kp_head_fast.hip (bf16, K <= 16) and kp_head_mfma.hip (C >= 128, K <= 64) against a torch fp32 computation on
    the same bf16-rounded activation."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N + H * 3 + C + K)
    a = torch.randn(N, H, W, C, generator=g).bfloat16()
    wt = torch.randn(K, C, generator=g) * C ** -0.5
    b = torch.randn(K, generator=g)
    gout = torch.randn(N, K, H, W, generator=g)
    af = a.float().requires_grad_(True)
    wr = wt.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    pre = torch.einsum('nhwc,kc->nkhw', af, wr) + br[None, :, None, None]
    # an output within rounding noise of the LeakyReLU kink may take either slope: no gradient through those
    gout = gout * (pre.detach().abs() > 1e-4)
    ref = torch.nn.functional.leaky_relu(pre, 0.01)
    ref.backward(gout)
    st = stream_ptr()
    ad, wd, bd, gd = a.cuda(), wt.cuda(), b.cuda(), gout.cuda()
    out = torch.full((N, K, H, W), float('nan'), device='cuda')
    call('kp_head_fwd', ptr(ad), ptr(wd), ptr(bd), 0.01, ptr(out), N, H, W, C, K, _lib.KP_BF16, st)
    assert (out.cpu() - ref.detach()).abs().max().item() < 1e-4 * ref.abs().max().item() + 1e-5
    dx = torch.full((N, H, W, C), float('nan'), device='cuda', dtype=torch.bfloat16)
    dw = torch.zeros(K, C, device='cuda'); db = torch.zeros(K, device='cuda')
    call('kp_head_bwd', ptr(gd), ptr(out), ptr(ad), ptr(wd), 0.01, ptr(dx), ptr(dw), ptr(db), None, N, H, W, C, K, _lib.KP_BF16, st)
    assert relerr(dx.float(), af.grad.numpy()) < 1e-2
    assert relerr(dw, wr.grad.numpy()) < 1e-4 and relerr(db, br.grad.numpy()) < 1e-4


@pytest.mark.parametrize('P,K,Cd,Cp,R', [(1, 64, 64, 64, 3), (8, 128, 256, 256, 3), (37, 64, 128, 128, 3), (5, 32, 48, 64, 3),
                                          (3, 16, 3, 3, 3), (256, 64, 3, 3, 3), (4, 64, 192, 192, 1), (6, 8, 20, 20, 3)])
def test_wgrad_unpack(P, K, Cd, Cp, R):
    """kp_wgrad_unpack: dw[k][c][r][s] += sum_p scratch[p][k][r][s][c] (LDS-transposing kernel for 64-multiples of
    the physical channel count, vector / scalar kernels otherwise; padded channels are dropped)."""
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(P * 7 + K + Cd)
    scratch = torch.randn(P, K, R, R, Cp, generator=g)
    dw0 = torch.randn(K, Cd, R, R, generator=g)
    ref = dw0.double() + scratch.double().sum(0)[..., :Cd].permute(0, 3, 1, 2)
    sd, dw = scratch.cuda(), dw0.cuda()
    call('kp_wgrad_unpack', ptr(sd), P, ptr(dw), K, Cd, Cp, R, stream_ptr())
    err = (dw.double().cpu() - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item() + 1e-6, err


def test_row_count_queries_match_the_launches():
    """kp_conv_fwd_stats_rows / kp_bn_bwd_rows (pure functions of the shape: nothing launched) return exactly the number
    of partial-sum rows the corresponding launch writes -- for every kernel family the dispatch can pick (LDS-patch
    persistent, linear implicit GEMM, first layer MFMA / VALU, small-channel, generic fp32; fast and general BN)."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, last_rows, ptr
    from keypoints_amd.runtime import precision_dtype, stream_ptr
    L = _lib.lib()
    st = stream_ptr()
    for (N, H, W, C, K, prec) in [(4, 32, 32, 64, 128, 'bf16'), (128, 16, 16, 512, 512, 'fp16'), (2, 24, 40, 64, 64, 'bf16'),
                                   (8, 128, 128, 3, 64, 'bf16'), (2, 20, 20, 3, 64, 'bf16'), (16, 84, 84, 16, 32, 'fp16'),
                                   (2, 16, 16, 5, 7, 'fp32'), (2, 16, 16, 64, 64, 'fp32')]:
        tdt, code = precision_dtype(prec)
        x = torch.randn(N, H + 2, W + 2, C, device='cuda').to(tdt)
        w = torch.randn(K, 3, 3, C, device='cuda').to(tdt)
        y = torch.empty(N, H, W, K, device='cuda', dtype=tdt)
        stats = torch.empty((N * H * W + 63) // 64, 2, K, device='cuda')
        q = L.kp_conv_fwd_stats_rows(N, H + 2, W + 2, C, K, 3, code)
        call('kp_conv_fwd', ptr(x), ptr(w), None, ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, code, st)
        assert 0 < q == last_rows() <= stats.shape[0], (N, H, W, C, K, prec, q, last_rows())
    for (N, H, W, C, tail, gp, prec) in [(8, 32, 32, 128, 0, 1, 'bf16'), (8, 32, 32, 128, 1, 1, 'fp16'), (2, 16, 16, 7, 0, 0, 'fp32'),
                                          (128, 64, 64, 256, 0, 0, 'bf16'), (3, 8, 8, 512, 1, 0, 'bf16')]:
        tdt, code = precision_dtype(prec)
        Hs, Ws = (H // 2, W // 2) if tail else (H, W)
        g = torch.randn(N, Hs + 2 * gp, Ws + 2 * gp, C, device='cuda').to(tdt)
        yv = torch.randn(N, H, W, C, device='cuda').to(tdt)
        am = torch.randint(0, 4, (N, Hs, Ws, C), device='cuda', dtype=torch.uint8) if tail else None
        vec = [torch.rand(C, device='cuda') + 0.5 for _ in range(4)]
        partials = torch.empty(N * H, 2, C, device='cuda')
        q = L.kp_bn_bwd_rows(gp, tail, N, H, W, C, code)
        call('kp_bn_act_bwd_reduce', ptr(g), gp, tail, ptr(am), ptr(yv), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]),
             0.01, ptr(partials), N, H, W, C, code, st)
        assert 0 < q == last_rows() <= N * H, (N, H, W, C, tail, gp, prec, q, last_rows())


@pytest.mark.parametrize('n,off,masked', [(4096 * 3, 0, False), (1001, 0, True), (4099, 1, False), (7, 0, True),
                                          (128 * 3 * 128 * 128, 0, False), (2 * 3 * 84 * 84 + 3, 2, True)])
def test_l2_loss_kernel(n, off, masked):
    """kp_l2_loss (transporter.py:56-60: mean((x_t - x_)^2 [* loss_mask]) and its gradient): 16-byte lanes with an
    element-wise remainder; unaligned pointers take the element-wise route throughout."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(n + off)
    x = torch.randn(n + off, generator=g).cuda()[off:]
    t = torch.randn(n + off, generator=g).cuda()[off:]
    m = (torch.rand(n + off, generator=g) > 0.3).float().cuda()[off:] if masked else None
    acc = torch.zeros(1, device='cuda', dtype=torch.float64)
    grad = torch.full((n + off,), float('nan'), device='cuda')[off:]
    call('kp_l2_loss', ptr(x), ptr(t), ptr(m), ptr(acc), ptr(grad), 1.0 / n, n, stream_ptr())
    d = (x.double() - t.double())
    mk = m.double() if masked else torch.ones_like(d)
    ref = (d * d * mk).sum()
    assert abs(acc.item() - ref.item()) <= 1e-6 * ref.item() + 1e-12
    gref = (2.0 * d * mk / n).float()
    assert torch.isfinite(grad).all()
    assert (grad - gref).abs().max().item() <= 1e-6 * gref.abs().max().item() + 1e-12


@pytest.mark.parametrize('dt', ['fp32', 'bf16'])
@pytest.mark.parametrize('N,C,Cp,H,W', [(3, 64, 64, 16, 16), (2, 10, 12, 7, 5), (2, 3, 4, 9, 11), (1, 5, 5, 4, 4)])
def test_nhwc_grad_to_nchw(N, C, Cp, H, W, dt):
    """kp_nhwc_grad_to_nchw: the adjoint of (replicate pad by 1, NCHW -> NHWC): border gradients fold onto the edge pixels.
    Both kernel routes (4 channels per thread when Cp % 4 == 0 and C > 4, else one pixel per thread) against autograd."""
    import torch.nn.functional as F
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(N + C + H)
    tdt = torch.float32 if dt == 'fp32' else torch.bfloat16
    gp = torch.randn(N, H + 2, W + 2, Cp, generator=g).to(tdt)
    x = torch.zeros(N, C, H, W, dtype=torch.float64, requires_grad=True)
    xp = F.pad(x, (1, 1, 1, 1), mode='replicate')
    (xp * gp[..., :C].double().permute(0, 3, 1, 2)).sum().backward()
    dx = torch.full((N, C, H, W), float('nan'), device='cuda')
    call('kp_nhwc_grad_to_nchw', ptr(gp.cuda()), ptr(dx), N, C, Cp, H, W, 1, _lib.KP_F32 if dt == 'fp32' else _lib.KP_BF16,
         stream_ptr())
    assert torch.isfinite(dx).all()
    assert (dx.cpu().double() - x.grad).abs().max().item() <= 1e-5


@pytest.mark.parametrize('dt', ['fp32', 'bf16'])
def test_pack_weights_batch_equals_per_layer_pack(dt):
    """kp_pack_weights_batch (all 3x3 layers of a Unit in one launch, LDS-tiled) == kp_pack_weights (one element per
    thread) layer by layer: forward pack [K][3][3][Cp] with zero padding channels, data-gradient pack [Cp][3][3][K] rotated
    by 180 degrees; ragged tiles (K, C not multiples of 32), a layer without a data-gradient pack."""
    import ctypes
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    code, tdt = (_lib.KP_F32, torch.float32) if dt == 'fp32' else (_lib.KP_BF16, torch.bfloat16)
    shapes = [(64, 3, 4, False), (128, 64, 64, True), (16, 16, 16, True), (32, 16, 16, True), (72, 40, 64, True), (512, 256, 256, True)]
    g = torch.Generator().manual_seed(5)
    st = stream_ptr()
    ws, wfs, wds, refs = [], [], [], []
    for K, C, Cp, has_d in shapes:
        w = torch.randn(K, C, 3, 3, generator=g).cuda()
        wf = torch.full((K, 3, 3, Cp), float('nan'), device='cuda', dtype=tdt)
        wd = torch.full((Cp, 3, 3, K), float('nan'), device='cuda', dtype=tdt) if has_d else None
        rf = torch.empty_like(wf); rd = torch.empty_like(wd) if has_d else None
        call('kp_pack_weights', ptr(w), ptr(rf), ptr(rd), K, C, Cp, 3, code, st)
        ws.append(w); wfs.append(wf); wds.append(wd); refs.append((rf, rd))
    n = len(shapes)
    vp, ia = ctypes.c_void_p * n, ctypes.c_int * n
    call('kp_pack_weights_batch', n, vp(*[ptr(t) for t in ws]), vp(*[ptr(t) for t in wfs]), vp(*[ptr(t) for t in wds]),
         ia(*[s[0] for s in shapes]), ia(*[s[1] for s in shapes]), ia(*[s[2] for s in shapes]), code, st)
    for (K, C, Cp, has_d), w, wf, wd, (rf, rd) in zip(shapes, ws, wfs, wds, refs):
        assert torch.equal(wf, rf), (K, C, Cp)
        if has_d:
            assert torch.equal(wd, rd), (K, C, Cp)
        assert torch.equal(wf[..., :C].float(), w.permute(0, 2, 3, 1).to(tdt).float())      # and the element kernel itself
        assert float(wf[..., C:].float().abs().sum()) == 0


@pytest.mark.parametrize('precision,tol', [('fp32', 2e-5), ('bf16', 3e-2), ('fp16', 4e-3)])
@pytest.mark.parametrize('tag,cin,cout,typ,enc', [('enc', 3, 5, 'VGG_PONG', True), ('dec', 16, 2, 'VGG_PONG', False),
                                                   ('pac', 1, 4, 'VGG_PACMAN', True)])
def test_unit_without_batchnorm_against_golden(golden, tag, cin, cout, typ, enc, precision, tol):
    """`knn.Unit(in, out, vgg.make_layers(cfg, batch_norm=False), batch_norm=False)` (knn.py:111-125, vgg.py:16-38) through the
    HIP engine against what the reference computed (tests/golden/nobn.npz): with no BatchNorm the conv bias is live (its
    gradient is the engine's `dbeta` slot), the activation pass runs with identity statistics, the backward's mean
    subtraction is off.  Max-pool and bilinear tails, LeakyReLU and ReLU cores, 1- / 3- / 16-channel inputs."""
    import torch.nn as nn
    from keypoints_amd.models import knn, vgg
    z = golden('nobn.npz')
    leaky = dict(nonlinearity=nn.LeakyReLU, nonlinearity_kwargs={'inplace': True}) if enc else {}
    cfg = (vgg.vgg_cfg if enc or tag == 'pac' else vgg.decoder_cfg)[typ]
    unit = knn.Unit(cin, cout, vgg.make_layers(cfg, batch_norm=False, **leaky), batch_norm=False)
    assert not any(isinstance(m, nn.BatchNorm2d) for m in unit.modules())
    sd = {}
    for blk in ('in_block', 'core', 'out_block'):
        for k in getattr(unit, blk).state_dict():
            sd[f'{blk}.{k}'] = torch.tensor(z[f'{tag}.p.{blk}.{k}'])
    unit.load_state_dict(sd, strict=True)          # same nn.Sequential keys as the reference's module
    unit = unit.cuda()
    unit.precision = precision
    x = torch.tensor(z[f'{tag}.x']).cuda().requires_grad_(True)
    y = unit(x)
    # fp32: every element to 2e-5 / 4e-5 of the tensor's maximum.  16-bit storage: the forward to the block tolerance; the
    # gradients per tensor in rel-L2 -- these Units are 3-5 layers WITHOUT a normalisation between them on 2 x (8 x 12 .. 20 x 20)
    # pixels: a rounding that flips one max-pool argmax or one ReLU gate moves single gradient entries by O(1) of their size
    # (measured max-abs / max up to 0.24 in bf16, 0.026 in fp16), so the element-wise bound of the one-block test does not apply
    def rl2(a, b):
        a, b = a.detach().cpu().double().numpy(), np.asarray(b, dtype=np.float64)
        return float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))
    gerr, gtol = (relerr, 2 * tol) if precision == 'fp32' else (rl2, dict(bf16=0.35, fp16=0.05)[precision])
    errs = {'y': (relerr(y, z[f'{tag}.y']), tol)}
    (y * torch.tensor(z[f'{tag}.w']).cuda()).sum().backward()
    errs['dx'] = (gerr(x.grad, z[f'{tag}.dx']), gtol)
    for blk in ('in_block', 'core', 'out_block'):
        for k, p in getattr(unit, blk).named_parameters():
            errs[f'{blk}.{k}'] = (gerr(p.grad, z[f'{tag}.g.{blk}.{k}']), gtol)
    print(f'nobn {tag} {precision}:', {k: f'{v[0]:.2e}' for k, v in errs.items()})
    assert all(e <= t for e, t in errs.values()), {k: v for k, v in errs.items() if v[0] > v[1]}


@pytest.mark.parametrize('dt', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('N,C,Cp,H,W,pad', [(3, 16, 16, 84, 84, 1), (2, 74, 128, 16, 16, 1), (2, 10, 16, 7, 5, 0), (2, 3, 3, 9, 11, 1),
                                           (1, 64, 64, 16, 16, 1), (2, 5, 8, 4, 6, 1)])
def test_nchw_to_nhwc(N, C, Cp, H, W, pad, dt):
    """Unit input layout: fp32 NCHW -> [N][H+2p][W+2p][Cp] in the storage type, replicate border (ReplicationPad2d(1),
    knn.py:115), zero channel padding -- the 8-channel-per-thread kernel (Cp % 8 == 0) and the per-pixel one, every element."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    tdt, code = {'fp32': (torch.float32, _lib.KP_F32), 'bf16': (torch.bfloat16, _lib.KP_BF16), 'fp16': (torch.float16, _lib.KP_F16)}[dt]
    g = torch.Generator().manual_seed(N + C + H)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    out = torch.full((N, H + 2 * pad, W + 2 * pad, Cp), float('nan'), device='cuda', dtype=tdt)
    call('kp_nchw_to_nhwc', ptr(x), ptr(out), N, C, Cp, H, W, pad, code, stream_ptr())
    ref = torch.nn.functional.pad(x, (pad,) * 4, mode='replicate') if pad else x
    ref = torch.nn.functional.pad(ref.permute(0, 2, 3, 1), (0, Cp - C)).to(tdt)
    assert torch.equal(out, ref)


@pytest.mark.parametrize('n,masked', [(4096 * 3, False), (1001, True), (7, True), (128 * 3 * 128 * 128, False), (2 * 3 * 84 * 84 + 3, True)])
def test_l2_loss_mean_kernel_one_launch_and_reusable_workspace(n, masked):
    """kp_l2_loss_mean: the loss VALUE of transporter.py:56-60 (mean in fp64, rounded once to fp32) written by the kernel's last
    workgroup, which also leaves the two-double workspace zero -- three calls in a row on one workspace give the same value."""
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g).cuda()
    t = torch.randn(n, generator=g).cuda()
    m = (torch.rand(n, generator=g) > 0.3).float().cuda() if masked else None
    ws = torch.zeros(2, device='cuda', dtype=torch.float64)
    d = (x.double() - t.double())
    mk = m.double() if masked else torch.ones_like(d)
    ref = float((d * d * mk).sum() / n)
    gref = (2.0 * d * mk / n).float()
    for _ in range(3):
        loss = torch.full((1,), float('nan'), device='cuda')
        grad = torch.full((n,), float('nan'), device='cuda')
        call('kp_l2_loss_mean', ptr(x), ptr(t), ptr(m), ptr(ws), ptr(loss), ptr(grad), n, stream_ptr())
        assert abs(loss.item() - ref) <= 2e-7 * abs(ref) + 1e-12, (loss.item(), ref)
        assert float(ws.abs().max()) == 0.0
        assert (grad - gref).abs().max().item() <= 1e-6 * gref.abs().max().item() + 1e-12


def test_l2_reconstruction_loss_matches_oracle_and_scales_its_gradient():
    """keypoints_amd.l2_reconstruction_loss (one launch forward, kp_scale backward) against the oracle, with an upstream factor."""
    import keypoints_amd as ka
    from oracle import kp_oracle as O
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 3, 20, 20, generator=g)
    t = torch.randn(3, 3, 20, 20, generator=g)
    m = (torch.rand(3, 3, 20, 20, generator=g) > 0.2).float()
    for mask in (None, m):
        xo = x.clone().double().requires_grad_(True)
        lo = O.l2_reconstruction_loss(xo, t.double(), None if mask is None else mask.double())
        (3.0 * lo).backward()
        xh = x.clone().cuda().requires_grad_(True)
        lh = ka.l2_reconstruction_loss(xh, t.cuda(), None if mask is None else mask.cuda())
        (3.0 * lh).backward()
        assert abs(lh.item() - lo.item()) <= 1e-6 * lo.item()
        assert float((xh.grad.cpu().double() - xo.grad).abs().max()) <= 1e-6 * float(xo.grad.abs().max())


@pytest.mark.parametrize('dt', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('N,C0,C1,Cp,H,W', [(3, 64, 10, 128, 16, 16), (2, 64, 10, 74, 16, 16), (2, 5, 3, 8, 7, 5), (1, 1, 1, 4, 4, 6)])
def test_two_source_layout_kernels_equal_the_concatenation(N, C0, C1, Cp, H, W, dt):
    """kp_nchw2_to_nhwc / kp_nhwc_grad_to_nchw2 (torch.cat((z, m), dim=1) of keynet.py:29 folded into the Unit's layout change):
    bit-identical to kp_nchw_to_nhwc / kp_nhwc_grad_to_nchw of the concatenated tensor, either output optional."""
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import precision_dtype, stream_ptr
    tdt, code = precision_dtype(dt)
    g = torch.Generator().manual_seed(N * 100 + C0)
    a = torch.randn(N, C0, H, W, generator=g).cuda()
    b = torch.randn(N, C1, H, W, generator=g).cuda()
    cat = torch.cat((a, b), dim=1).contiguous()
    st = stream_ptr()
    for pad in (0, 1):
        want = torch.empty(N, H + 2 * pad, W + 2 * pad, Cp, device='cuda', dtype=tdt)
        got = torch.full_like(want, float('nan'))
        call('kp_nchw_to_nhwc', ptr(cat), ptr(want), N, C0 + C1, Cp, H, W, pad, code, st)
        call('kp_nchw2_to_nhwc', ptr(a), C0, ptr(b), C1, ptr(got), N, Cp, H, W, pad, code, st)
        assert torch.equal(got.float(), want.float())
        gsrc = torch.randn(N, H + 2 * pad, W + 2 * pad, Cp, generator=g).cuda().to(tdt)
        dcat = torch.empty(N, C0 + C1, H, W, device='cuda')
        call('kp_nhwc_grad_to_nchw', ptr(gsrc), ptr(dcat), N, C0 + C1, Cp, H, W, pad, code, st)
        d0 = torch.full((N, C0, H, W), float('nan'), device='cuda')
        d1 = torch.full((N, C1, H, W), float('nan'), device='cuda')
        call('kp_nhwc_grad_to_nchw2', ptr(gsrc), ptr(d0), C0, ptr(d1), C1, N, Cp, H, W, pad, code, st)
        assert torch.equal(d0, dcat[:, :C0]) and torch.equal(d1, dcat[:, C0:])
        d1b = torch.full((N, C1, H, W), float('nan'), device='cuda')
        call('kp_nhwc_grad_to_nchw2', ptr(gsrc), None, C0, ptr(d1b), C1, N, Cp, H, W, pad, code, st)
        assert torch.equal(d1b, d1)


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'fp16'])
def test_unit_forward_cat_equals_cat_then_forward(precision):
    """Unit.forward_cat(z, m) == Unit(torch.cat((z, m), dim=1)) (KeyNet's decoder call, keynet.py:29): outputs, both input
    gradients and every parameter gradient bit for bit."""
    from keypoints_amd.models import knn, vgg
    torch.manual_seed(11)
    unit = knn.Unit(64 + 10, 3, vgg.make_layers([64, 64, 'U', 64])).cuda()
    unit.precision = precision
    g = torch.Generator().manual_seed(2)
    z0 = torch.randn(4, 64, 16, 16, generator=g).cuda()
    m0 = torch.rand(4, 10, 16, 16, generator=g).cuda()
    w = torch.randn(4, 3, 32, 32, generator=g).cuda()
    res = []
    for cat in (True, False):
        z, m = z0.clone().requires_grad_(True), m0.clone().requires_grad_(True)
        for p in unit.parameters():
            p.grad = None
        unit._arena = None
        out = unit(torch.cat((z, m), dim=1)) if cat else unit.forward_cat(z, m)
        (out * w).sum().backward()
        res.append((out.detach().clone(), z.grad.clone(), m.grad.clone(), [p.grad.clone() for p in unit.parameters()]))
    (o1, z1, m1, p1), (o2, z2, m2, p2) = res
    assert torch.equal(o1, o2) and torch.equal(z1, z2) and torch.equal(m1, m2)
    for a, b in zip(p1, p2):
        assert torch.equal(a, b)


@pytest.mark.parametrize('core,slope_core', [([16, 32], 'leaky'), ([64, 'M', 64, 'U', 64], 'relu')])
def test_backward_through_eval_mode_batchnorm(core, slope_core):
    """`net.eval()` + a backward (fine-tuning with frozen statistics: nn.BatchNorm2d in eval mode is an affine map with constant
    running statistics): the Unit's forward and every gradient -- input, conv weights AND conv biases (non-zero here, unlike in front
    of train-mode BatchNorm), gamma, beta -- against the oracle's eval-mode Unit under torch autograd, strict fp32."""
    import torch.nn as nn
    from keypoints_amd.models import knn, vgg
    from oracle import kp_oracle as O
    torch.manual_seed(5)
    kw = dict(nonlinearity=nn.LeakyReLU, nonlinearity_kwargs={'inplace': True}) if slope_core == 'leaky' else {}
    cin, cout = 3, 5
    unit = knn.Unit(cin, cout, vgg.make_layers(core, **kw))
    for m in unit.modules():      # non-trivial parameters and running statistics
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu'); m.bias.data.normal_(0, 0.1)
    plan = O.unit_plan(core, cin, cout, 0.01 if slope_core == 'leaky' else 0.0)
    P = {f'u.{n}': p.detach().clone().double().requires_grad_(True) for n, p in unit.named_parameters()}
    Bf = {f'u.{n}': b.detach().clone().double() if b.dtype.is_floating_point else b.detach().clone() for n, b in unit.named_buffers()}
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, cin, 16, 16, generator=g)
    xo = x.double().requires_grad_(True)
    out_o = O.unit_forward(plan, P, 'u', xo, Bf, training=False)
    w = torch.randn(out_o.shape, generator=g)
    (out_o * w.double()).sum().backward()
    unit = unit.cuda().eval()
    before = {n: b.detach().clone() for n, b in unit.named_buffers()}
    xh = x.cuda().requires_grad_(True)
    out = unit(xh)
    (out * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    close = lambda a, b, tol: float((a.detach().cpu().double() - b.detach()).abs().max()) <= tol * max(1e-6, float(b.detach().abs().max()))
    assert close(out, out_o, 2e-5)
    assert close(xh.grad, xo.grad, 1e-4)
    for n, p in unit.named_parameters():
        assert close(p.grad, P[f'u.{n}'].grad, 2e-4), n
        assert float(p.grad.abs().max()) > 0, n          # conv biases included: eval-mode BatchNorm does not cancel them
    for n, b in unit.named_buffers():                    # eval mode leaves the running statistics alone
        assert torch.equal(b, before[n]), n


@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
def test_second_backward_with_kept_activations(precision):
    """`backward(retain_graph=True)` twice through one forward (runtime.set_keep_saved_activations): the second pass accumulates the
    same gradients again -- `.grad` exactly doubles (fp32 g + g), the input gradient is bit-identical -- and without the switch the
    second backward raises instead of reading released buffers."""
    from keypoints_amd import runtime
    from keypoints_amd.models import knn, vgg
    torch.manual_seed(3)
    unit = knn.Unit(3, 5, vgg.make_layers([64, 'M', 64, 'U', 64])).cuda()
    unit.precision = precision
    x = torch.randn(4, 3, 32, 32, device='cuda', requires_grad=True)
    w = torch.randn(4, 5, 32, 32, device='cuda')
    runtime.set_keep_saved_activations(True)
    try:
        loss = (unit(x) * w).sum()
        loss.backward(retain_graph=True)
        g1 = [p.grad.clone() for p in unit.parameters()]
        x1 = x.grad.clone()
        x.grad = None
        loss.backward()
        for a, (name, p) in zip(g1, unit.named_parameters()):
            assert torch.equal(p.grad, a + a), name
            assert p.dim() == 1 or float(a.abs().max()) > 0, name   # (conv biases in front of train-mode BatchNorm: zero gradient)
        assert torch.equal(x.grad, x1)
    finally:
        runtime.set_keep_saved_activations(False)
    loss = (unit(x) * w).sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match='set_keep_saved_activations'):
        loss.backward()
