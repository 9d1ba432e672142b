"""Winograd F(2, 3) form of the 3x3 layers (kp_conv_wino.hip, fp16 storage) through the C ABI.

Checker 1 (bit-level contract of the kernel): a torch fp32 restatement of exactly what the kernel computes -- row
differences / sums of the fp16 input re-rounded to fp16, filter-row combinations of the fp16 weights re-rounded to fp16 (the
kernel's own kp_conv_wino_pack is compared with it bit for bit), products accumulated in fp32, output transform in fp32 -- :
every output element within one fp16 ulp (+ fp32 accumulation-order noise).
Checker 2 (what the reference computes): nn.Conv2d(3x3) after ReplicationPad2d(1) (keypoints/models/vgg.py:30-33) on the same
fp16 operands in fp32.  STATED TOLERANCE of the Winograd form against it: rel-L2 <= 1e-3, every element
within 4 fp16 ulps of |y| + 1e-3 max|y| -- the direct kernels meet 1 ulp; the end-to-end contract (keypoints 5e-3 on the
conditioned model F) is asserted in tests/test_gpu_fullsize.py."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

KP_CFG_WALK, KP_CFG_CUS = 5, 6
ULP = 2.0 ** -10
h = lambda t: t.half().float()


@pytest.fixture(autouse=True)
def _knobs():
    from keypoints_amd._lib import call
    yield
    call('kp_config', KP_CFG_CUS, 0)
    call('kp_config', KP_CFG_WALK, 1 | 2 | 4 | 128 | 0xfe00)


def wino_pack_ref(wf):
    """wf [K][3 r][3 s][C] fp16 -> U [K][4][3][C] fp16 (G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] over the filter rows)."""
    g = wf.float()
    g0, g1, g2 = g[:, 0], g[:, 1], g[:, 2]
    return torch.stack((g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2), dim=1).half()


def wino_ref(x, u, bias):
    """x [N][H+2][W+2][C] fp16, u [K][4][3][C] fp16 -> y [N][H][W][K] fp32: the kernel's arithmetic in torch."""
    d = x.float()
    H = x.shape[1] - 2
    dj = [d[:, j:j + H:2] for j in range(4)]                       # input rows 2t + j of every row pair t
    V = [h(dj[0] - dj[2]), h(dj[1] + dj[2]), h(dj[2] - dj[1]), h(dj[1] - dj[3])]
    M = [F.conv2d(V[p].permute(0, 3, 1, 2), u[:, p].float().permute(0, 2, 1)[:, :, None, :]) for p in range(4)]   # [N][K][H/2][W]
    y0, y1 = M[0] + M[1] + M[2], M[1] - M[2] - M[3]
    y = torch.stack((y0, y1), dim=3).reshape(y0.shape[0], y0.shape[1], H, -1)
    if bias is not None:
        y = y + bias[None, :, None, None]
    return y.permute(0, 2, 3, 1).contiguous()


FWD = [  # N, H, W, C, K, CU budget, linear walk.  (64 output channels per workgroup tile: K / 64 k-tiles; a workgroup that walks
    # several tiles must keep ONE k-tile for its statistics rows: grid / 8 a multiple of the k-tile count)
    (2, 16, 16, 64, 64, 8, 1), (2, 32, 32, 128, 256, 16, 1), (6, 16, 16, 64, 256, 16, 1), (8, 48, 48, 64, 128, 16, 1),
    (8, 48, 48, 64, 128, 16, 0), (8, 32, 64, 256, 256, 32, 1), (8, 32, 64, 96, 64, 16, 0), (4, 16, 16, 512, 512, 256, 1),
]


@pytest.mark.parametrize('N,H,W,C,K,cus,linear', FWD)
def test_wino_forward_and_statistics(N, H, W, C, K, cus, linear):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    code = _lib.KP_F16
    g = torch.Generator().manual_seed(N * 1000 + C + K + H)
    x = torch.randn(N, H + 2, W + 2, C, generator=g).half()
    w = (torch.randn(K, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).half()
    bias = torch.randn(K, generator=g)
    st = stream_ptr()
    call('kp_config', KP_CFG_WALK, 2 | 4 | 128 | 0xfe00 | linear)
    call('kp_config', KP_CFG_CUS, cus)
    assert _lib.lib().kp_conv_wino_supported(N, H, W, C, K, 1, code) == 1
    xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
    u = torch.full((K, 4, 3, C), float('nan'), device='cuda', dtype=torch.float16)
    call('kp_conv_wino_pack', ptr(wd), ptr(u), K, C, code, st)
    uref = wino_pack_ref(w)
    assert torch.equal(u.cpu(), uref)
    rows = _lib.lib().kp_conv_fwd_wino_rows(N, H + 2, W + 2, C, K, code)
    assert rows > 0
    y = torch.full((N, H, W, K), float('nan'), device='cuda', dtype=torch.float16)
    stats = torch.full((rows, 2, K), float('nan'), device='cuda')
    call('kp_conv_fwd_wino', ptr(xd), ptr(u), ptr(bd), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, code, st)
    got = y.float().cpu()
    assert torch.isfinite(got).all()
    emu = wino_ref(x, uref, bias)
    tol = ULP * emu.abs() + 2e-5 * emu.abs().max()
    bad = (got - emu).abs() > tol
    assert not bad.any(), (int(bad.sum()), float((got - emu).abs().max()))
    # against the convolution the reference computes
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias).permute(0, 2, 3, 1)
    rel = float((got - ref).norm() / ref.norm())
    worst = float(((got - ref).abs() - 4 * ULP * ref.abs()).max() / ref.abs().max())
    print(f'wino vs conv: rel-L2 {rel:.2e}, worst beyond 4 ulp {worst:.2e} of max|y|')
    assert rel <= 1e-3 and worst <= 1e-3
    # statistics: sums of the fp32 outputs (before the storage rounding)
    s = stats.double().sum(0).cpu()
    flat = emu.double().reshape(-1, K)
    assert torch.isfinite(s).all()
    assert (s[0] - flat.sum(0)).abs().max().item() <= 1e-4 * flat.abs().sum(0).max().item()
    assert (s[1] - (flat ** 2).sum(0)).abs().max().item() <= 1e-4 * (flat ** 2).sum(0).max().item()
    # bitwise repeatable
    y2 = torch.full_like(y, float('nan'))
    stats2 = torch.full_like(stats, float('nan'))
    call('kp_conv_fwd_wino', ptr(xd), ptr(u), ptr(bd), ptr(y2), ptr(stats2), N, H + 2, W + 2, C, K, code, st)
    assert torch.equal(y, y2) and torch.equal(stats, stats2)


DGRAD = [(4, 32, 32, 128, 256, 8), (6, 16, 16, 512, 64, 8), (2, 48, 48, 64, 128, 8), (8, 16, 32, 256, 128, 16), (4, 64, 64, 128, 64, 256)]


@pytest.mark.parametrize('N,H,W,K,C,cus', DGRAD)
def test_wino_data_gradient(N, H, W, K, C, cus):
    """gx over the padded domain [H+2][W+2]: interior by the Winograd kernel (offset (1, 1)), 3-tap ring by the direct kernel."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    code = _lib.KP_F16
    g = torch.Generator().manual_seed(7 * N + K + C + W)
    dyi = torch.randn(N, H, W, K, generator=g).half()
    wd = (torch.randn(C, 3, 3, K, generator=g) * (2.0 / (9 * K)) ** 0.5).half()    # the flipped pack [C][3][3][K]
    dy = torch.zeros(N, H + 4, W + 4, K, dtype=torch.float16)
    dy[:, 2:-2, 2:-2] = dyi
    ref = F.conv2d(dy.float().permute(0, 3, 1, 2), wd.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    st = stream_ptr()
    call('kp_config', KP_CFG_CUS, cus)
    dyd, wdd = dy.cuda(), wd.cuda()
    ud = torch.empty(C, 4, 3, K, device='cuda', dtype=torch.float16)
    call('kp_conv_wino_pack', ptr(wdd), ptr(ud), C, K, code, st)
    gx = torch.full((N, H + 2, W + 2, C), float('nan'), device='cuda', dtype=torch.float16)
    call('kp_conv_dgrad_wino', ptr(dyd), ptr(ud), ptr(wdd), ptr(gx), N, H, W, K, C, code, st)
    gd = torch.full((N, H + 2, W + 2, C), float('nan'), device='cuda', dtype=torch.float16)
    call('kp_conv_dgrad', ptr(dyd), ptr(wdd), ptr(gd), N, H, W, K, C, code, st)
    got, direct = gx.float().cpu(), gd.float().cpu()
    assert torch.isfinite(got).all()
    emu = wino_ref(dy[:, 1:-1, 1:-1], wino_pack_ref(wd), None)     # the interior: rows / columns 1 .. H / W of the padded domain
    inner = got[:, 1:-1, 1:-1]
    tol = ULP * emu.abs() + 2e-5 * emu.abs().max()
    assert not ((inner - emu).abs() > tol).any(), float((inner - emu).abs().max())
    rel = float((got - ref).norm() / ref.norm())
    assert rel <= 1e-3, rel
    ring = torch.ones(H + 2, W + 2, dtype=torch.bool)
    ring[1:-1, 1:-1] = False
    assert torch.equal(got[:, ring], direct[:, ring])              # the ring is the direct kernel's, bit for bit
