"""Row f3 widened to the matrix-core class: keypoints_amd.inference.FusedUnitF -- the eval-mode forward of model F's
keypoint Unit (/root/reference/atari_demo.py:35-36 runs `net.keypoint(x)` of a trained net with BatchNorm in eval mode) with every
BatchNorm + activation in the epilogue of its convolution (kp_conv_fwd_act) -- against the oracle's eval-mode Unit in fp64 on the SAME
16-bit-rounded weights, next to the oracle's own emulation of 16-bit storage (what the layer chain computes), and the op itself
against torch at one unit in the last place of the storage type."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ULP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}   # half a unit in the last place, relative


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('N,H,W,C,K,pad,slope', [(4, 32, 32, 128, 256, 1, 0.01), (2, 64, 64, 64, 128, 0, 0.0), (3, 16, 48, 256, 64, 1, 0.2),
                                                (130, 16, 16, 64, 64, 1, 0.01)])
def test_conv_fwd_act_one_rounding(N, H, W, C, K, pad, slope, dt):
    """kp_conv_fwd_act: act(conv * scale + shift) from the fp32 accumulators, rounded ONCE, into the interior of a padded buffer whose
    ring kp_pad_border then fills -- every element against an fp64 evaluation of the same rounded operands, NaN-prefilled output."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    code = _lib.KP_F16 if dt == torch.float16 else _lib.KP_BF16
    g = torch.Generator().manual_seed(N + H + C + K)
    x = torch.randn(N, H + 2, W + 2, C, generator=g).to(dt)
    w = (torch.randn(K, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dt)
    scale = torch.rand(K, generator=g) + 0.5
    shift = torch.randn(K, generator=g)
    assert _lib.lib().kp_conv_fwd_act_supported(N, H, W, C, K, code) == 1
    conv = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    z = conv * scale.double() + shift.double()
    ref = torch.maximum(z, z * slope)
    xd, wd, sc, sh = x.cuda(), w.cuda(), scale.cuda(), shift.cuda()
    out = torch.full((N, H + 2 * pad, W + 2 * pad, K), float('nan'), device='cuda', dtype=dt)
    call('kp_conv_fwd_act', ptr(xd), ptr(wd), ptr(sc), ptr(sh), slope, ptr(out), pad, N, H, W, C, K, code, stream_ptr())
    if pad:
        assert torch.isnan(out[:, 0].float()).all()                       # the ring is not the convolution's to write
        call('kp_pad_border', ptr(out), N, H, W, K, pad, code, stream_ptr())
    torch.cuda.synchronize()
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    inner = got[:, pad:pad + H, pad:pad + W] if pad else got
    tol = 2 * ULP[dt] * ref.abs() + 3e-6 * ref.abs().max()               # one rounding + fp32 accumulation of 9 C products
    assert ((inner - ref).abs() <= tol).all(), float(((inner - ref).abs() - tol).max())
    if pad:
        want = F.pad(inner.permute(0, 3, 1, 2), (pad,) * 4, mode='replicate').permute(0, 2, 3, 1)
        assert torch.equal(got, want)


@pytest.mark.parametrize('precision', ['fp16', 'bf16'])
def test_fused_unit_F_against_the_oracle_eval_unit(precision):
    """Model F's keypoint Unit in eval mode, batch 4: FusedUnitF's heat-maps against the fp64 oracle (`unit_forward(training=False)`) on
    the same 16-bit-rounded conv weights.  Fewer roundings than the layer chain (one per layer instead of two): the fused path must
    be no further from fp64 than the oracle's emulation of the chain's 16-bit storage is (x 1.25), and the keypoints agree with the
    chain's to the storage type's resolution."""
    from keypoints_amd.inference import FusedUnitF, KeypointExtractor
    from keypoints_amd.models import functional as MF, transporter
    from oracle import kp_oracle as O
    torch.manual_seed(5)
    net = transporter.make('F', 3, 64, 10)
    g = torch.Generator().manual_seed(11)
    for name, buf in net.named_buffers():
        if name.endswith('running_mean'):
            buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
        elif name.endswith('running_var'):
            buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
    x = torch.rand(4, 3, 128, 128, generator=g)
    plans = O.transporter_plans('F', 3, 64, 10)
    P = {k: v.detach().clone() for k, v in net.named_parameters()}
    Bf = {k: v.detach().clone() for k, v in net.named_buffers()}
    quant = O.fp16_storage if precision == 'fp16' else O.bf16_storage
    P64 = {k: v.double() for k, v in P.items()}
    B64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in Bf.items()}
    B32 = {k: v.clone() for k, v in Bf.items()}
    kplan = plans['keypoint']
    with torch.no_grad():
        ref = O.unit_forward(kplan, P64, 'keypoint', x.double(), buffers=B64, training=False, quant=None)
        emu = O.unit_forward(kplan, P, 'keypoint', x, buffers=B32, training=False, quant=quant)
    net = net.cuda().set_precision(precision).eval()
    fu = FusedUnitF(net.keypoint)
    heat = fu(x.cuda())
    assert fu.fused_layers == 8 and heat.shape == ref.shape      # all eight 3x3 layers of the Unit
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
    e_fused, e_emu = rel(heat, ref), rel(emu, ref)
    print(f'FusedUnitF {precision}: rel-L2 to the fp64 oracle {e_fused:.3e}; the oracle with {precision} storage emulated {e_emu:.3e}')
    assert e_fused <= 1.25 * e_emu + 1e-4, (e_fused, e_emu)
    with torch.no_grad():
        chain = net.keypoint(x.cuda())
    k_f, k_c = MF.spacial_logsoftmax(heat), MF.spacial_logsoftmax(chain)
    k_r = O.spatial_logsoftmax(ref.float())[0]
    tol = 5e-3 if precision == 'fp16' else 8e-2       # oracle.TOL16's keypoint clauses
    assert float((k_f.cpu() - k_r).abs().max()) <= tol and float((k_c.cpu() - k_r).abs().max()) <= tol
    # the extractor picks it, eager and as a HIP graph, bit for bit
    ex_g, ex_e = KeypointExtractor(net, graph=True, fused=True), KeypointExtractor(net, graph=False, fused=True)
    assert torch.equal(ex_g(x.cuda()), ex_e(x.cuda())) and torch.equal(ex_e(x.cuda()), k_f)
    # a strict-fp32 network has no such form: fused=True says so, 'auto' takes the chain
    net32 = transporter.make('F', 3, 64, 10).cuda().eval()
    with pytest.raises(NotImplementedError):
        KeypointExtractor(net32, fused=True)
    assert KeypointExtractor(net32, graph=False)._fused_unit is None
