"""Second-generation LDS-patch convolution (kp_conv_pair.hip: two 16x16 pixel tiles per workgroup, 128 x 64 wave tile,
32-channel chunks) through the C ABI: forward with BatchNorm statistics, and the data gradient over the padded domain,
against a torch fp32 convolution of the same 16-bit-rounded operands (nn.Conv2d after ReplicationPad2d(1), vgg.py:30-33;
the pad is physically present in the input) and against the first-generation kernel it replaces.

The CU budget knob (KP_CFG_CUS) shrinks the persistent grid so that workgroups walk SEVERAL tiles (tile switch, bias reload,
k-tile carries, statistics accumulated over tiles) at sizes a CPU reference finishes in seconds; odd tiles-per-row counts make
the second pixel tile of a pair the first tile of the next row / image."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

KP_CFG_CONV_PATCH, KP_CFG_WALK, KP_CFG_CUS = 0, 5, 6
ULP = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}


@pytest.fixture(autouse=True)
def _knobs():
    from keypoints_amd._lib import call, lib
    walk = 1 | 2 | 4 | 128 | 0xfe00   # KP_WALK_DEFAULT
    yield
    call('kp_config', KP_CFG_CONV_PATCH, 1)
    call('kp_config', KP_CFG_CUS, 0)
    call('kp_config', KP_CFG_WALK, walk)


def _pair_grid(N, H, W, K, cus):
    ptiles = N * (H // 16) * (W // 16)
    ktiles = K // (128 if K % 128 == 0 else 64)
    ntiles = ptiles // 2 * ktiles
    return min((cus // 8) * 8, (ntiles + 7) // 8 * 8), ntiles, ktiles


FWD = [  # N, H, W, C, K, CU budget, linear walk
    (2, 32, 32, 128, 256, 8, 1),     # 8 tiles on 8 workgroups: one tile each, two k-tiles
    (8, 16, 16, 64, 256, 8, 1),      # one tile per image: the pair spans two images
    (16, 16, 16, 128, 128, 8, 1),
    (8, 48, 48, 64, 128, 8, 1),      # 3 tiles per row: pairs wrap rows and images; 36 tiles on 8 workgroups (ragged)
    (8, 48, 48, 64, 128, 8, 0),      # ... in 8 per-XCD ranges
    (8, 32, 64, 256, 256, 16, 1),    # 64 tiles on 16 workgroups, k-tile fastest
    (8, 32, 64, 256, 256, 16, 0),
    (4, 32, 32, 128, 64, 8, 1),      # 64-channel k-tile
    (8, 16, 16, 512, 256, 256, 1),   # the bottleneck-side maps, whole device (statistics: 4 rows per workgroup must fit the N H W / 64 rows of the caller's buffer)
]


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('N,H,W,C,K,cus,linear', FWD)
def test_pair_forward_and_statistics(N, H, W, C, K, cus, linear, dt):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    code = _lib.KP_F16 if dt == torch.float16 else _lib.KP_BF16
    g = torch.Generator().manual_seed(N * 1000 + C + K + H)
    x = torch.randn(N, H + 2, W + 2, C, generator=g).to(dt)
    w = (torch.randn(K, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dt)
    bias = torch.randn(K, generator=g)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias).permute(0, 2, 3, 1).contiguous()
    xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
    st = stream_ptr()
    walk = 2 | 4 | 128 | 0xfe00
    call('kp_config', KP_CFG_WALK, walk | linear)
    call('kp_config', KP_CFG_CUS, cus)
    out = {}
    for mode in (2, 1):
        call('kp_config', KP_CFG_CONV_PATCH, mode)
        y = torch.full((N, H, W, K), float('nan'), device='cuda', dtype=dt)
        stats = torch.full(((N * H * W + 63) // 64, 2, K), float('nan'), device='cuda')
        call('kp_conv_fwd', ptr(xd), ptr(wd), ptr(bd), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, code, st)
        rows = _lib.last_rows()
        assert rows == _lib.lib().kp_conv_fwd_stats_rows(N, H + 2, W + 2, C, K, 3, code)
        out[mode] = (y.float().cpu(), stats[:rows].double().sum(0).cpu(), rows)
    grid, ntiles, ktiles = _pair_grid(N, H, W, K, cus)
    assert out[2][2] == 4 * grid, 'the pair kernel did not take this shape'
    got, s, _ = out[2]
    assert torch.isfinite(got).all()
    # every element within one unit in the last place of the storage type of the fp32 result (+ accumulation-order noise)
    tol = ULP[dt] * ref.abs() + 2e-5 * ref.abs().max()
    bad = ((got - ref).abs() > tol)
    assert not bad.any(), (int(bad.sum()), float((got - ref).abs().max()))
    # and within one ulp of what the first-generation kernel stores (another order of the same fp32 accumulation)
    assert ((got - out[1][0]).abs() <= ULP[dt] * ref.abs() + 2e-5 * ref.abs().max()).all()
    # statistics: sums of the fp32 accumulators.  Against the first-generation kernel (the same accumulators up to the order
    # of the additions) 2e-5 of the per-channel total; against the fp64 sums of the reference 1e-4 (the matrix cores
    # accumulate 9 C products per pixel with their own rounding, a common offset of ~1e-5 |y| at C = 512)
    flat = ref.double().reshape(-1, K)
    assert torch.isfinite(s).all()
    s1 = out[1][1]
    assert (s[0] - s1[0]).abs().max().item() <= 2e-5 * flat.abs().sum(0).max().item()
    assert (s[1] - s1[1]).abs().max().item() <= 2e-5 * (flat ** 2).sum(0).max().item()
    assert (s[0] - flat.sum(0)).abs().max().item() <= 1e-4 * flat.abs().sum(0).max().item()
    assert (s[1] - (flat ** 2).sum(0)).abs().max().item() <= 1e-4 * (flat ** 2).sum(0).max().item()
    # the same launch twice: bitwise the same output and statistics
    y2 = torch.full((N, H, W, K), float('nan'), device='cuda', dtype=dt)
    stats2 = torch.full(((N * H * W + 63) // 64, 2, K), float('nan'), device='cuda')
    call('kp_config', KP_CFG_CONV_PATCH, 2)
    call('kp_conv_fwd', ptr(xd), ptr(wd), ptr(bd), ptr(y2), ptr(stats2), N, H + 2, W + 2, C, K, 3, code, st)
    call('kp_conv_fwd', ptr(xd), ptr(wd), ptr(bd), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, code, st)
    assert torch.equal(y, y2) and torch.equal(stats[:4 * grid], stats2[:4 * grid])


DGRAD = [  # N, H, W, K (gradient channels), C (input channels), CU budget
    (4, 32, 32, 128, 256, 8), (6, 16, 16, 512, 64, 8), (2, 48, 48, 64, 128, 8), (8, 16, 32, 256, 128, 16),
    (4, 64, 64, 128, 64, 256),
]


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('N,H,W,K,C,cus', DGRAD)
def test_pair_data_gradient(N, H, W, K, C, cus, dt):
    """gx over the padded domain [H+2][W+2] = 16x16-tileable interior (pair kernel, offset (1, 1)) + 3-tap border ring."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    code = _lib.KP_F16 if dt == torch.float16 else _lib.KP_BF16
    g = torch.Generator().manual_seed(7 * N + K + C + W)
    dyi = torch.randn(N, H, W, K, generator=g).to(dt)
    wd = (torch.randn(C, 3, 3, K, generator=g) * (2.0 / (9 * K)) ** 0.5).to(dt)    # the flipped pack [C][3][3][K]
    dy = torch.zeros(N, H + 4, W + 4, K, dtype=dt)
    dy[:, 2:-2, 2:-2] = dyi
    ref = F.conv2d(dy.float().permute(0, 3, 1, 2), wd.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    dyd, wdd = dy.cuda(), wd.cuda()
    st = stream_ptr()
    call('kp_config', KP_CFG_CUS, cus)
    res = {}
    for mode in (2, 1):
        call('kp_config', KP_CFG_CONV_PATCH, mode)
        gx = torch.full((N, H + 2, W + 2, C), float('nan'), device='cuda', dtype=dt)
        call('kp_conv_dgrad', ptr(dyd), ptr(wdd), ptr(gx), N, H, W, K, C, code, st)
        res[mode] = gx.float().cpu()
    got = res[2]
    assert torch.isfinite(got).all()
    tol = ULP[dt] * ref.abs() + 2e-5 * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bad.any(), (int(bad.sum()), float((got - ref).abs().max()))
    assert ((got - res[1]).abs() <= tol).all()
    # the ring is the same kernel in both modes: bit-identical there
    ring = torch.ones(H + 2, W + 2, dtype=torch.bool)
    ring[1:-1, 1:-1] = False
    assert torch.equal(got[:, ring], res[1][:, ring])


def test_pair_declines_what_it_cannot_tile():
    """An odd number of pixel tiles has no pairing: the first-generation kernel takes the shape (same results as before)."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    N, H, W, C, K = 3, 16, 16, 64, 128
    x = torch.randn(N, H + 2, W + 2, C).half().cuda()
    w = (torch.randn(K, 3, 3, C) * 0.05).half().cuda()
    outs = []
    for mode in (2, 1):
        call('kp_config', KP_CFG_CONV_PATCH, mode)
        y = torch.empty(N, H, W, K, device='cuda', dtype=torch.float16)
        stats = torch.empty(N * H * W // 64, 2, K, device='cuda')
        call('kp_conv_fwd', ptr(x), ptr(w), None, ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, _lib.KP_F16, stream_ptr())
        outs.append((y, stats[:_lib.last_rows()].clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
