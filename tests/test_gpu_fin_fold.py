"""kp_conv_fwd_fin (include/kp_hip.h): Conv2d(3x3) + the BatchNorm2d statistics of vgg.py:33-35 as ONE call.  On the persistent patch
kernel the finalize runs in the tail of the convolution (the last workgroup of each output-channel tile, device-scope tickets);
everywhere else the call launches kp_bn_finalize itself.  Against the two separate calls: the convolution output and the statistics
rows bit for bit, mean / invstd / running statistics to one unit in the last place of fp32 (another fixed order of the same fp64
sums), num_batches_tracked incremented once, the ticket counters zero again -- also when two such launches run on two streams at
the same time, and launch after launch on one stream."""
import pytest
import torch

pytestmark = pytest.mark.gpu

KP_CFG_FIN_FOLD = 16

SHAPES = [  # N, H, W, C, K, folds (on a 256-CU device)
    (4, 128, 128, 64, 128, True),     # one k-tile of 128, 256 pixel tiles = one per workgroup
    (16, 64, 64, 128, 256, True),     # two k-tiles
    (64, 32, 32, 256, 512, True),     # four k-tiles
    (8, 128, 128, 128, 64, True),     # the 64-channel tile (BN = 64)
    (5, 128, 128, 64, 128, True),     # 320 pixel tiles over 256 workgroups: some workgroups run two tiles
    (2, 64, 64, 64, 128, True),       # 32 pixel tiles: a grid of 32 workgroups, one tile each
    (3, 48, 48, 64, 128, False),      # 27 tiles on a grid of 32: per-tile rows (not one row per workgroup), finalize as its own launch
    (8, 40, 40, 64, 64, False),       # not 16 x 16 tileable: the linear kernel
    (8, 84, 84, 16, 32, False),       # a Pong-class layer
]


def _ulp_close(a, b, n=1):
    return bool(((a - b).abs() <= n * torch.finfo(torch.float32).eps * b.abs().clamp_min(1e-30)).all())


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('N,H,W,C,K,folds', SHAPES)
def test_conv_fwd_fin_equals_the_two_calls(N, H, W, C, K, folds, dt):
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    L = _lib.lib()
    code = _lib.KP_F16 if dt == torch.float16 else _lib.KP_BF16
    g = torch.Generator().manual_seed(N * 1000 + H + C + K)
    x = torch.randn(N, H + 2, W + 2, C, generator=g).cuda().to(dt)
    w = (torch.randn(K, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda().to(dt)
    bias = torch.randn(K, generator=g).cuda()
    st = stream_ptr()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if cus == 256:
        assert L.kp_conv_fwd_fin_folds(N, H + 2, W + 2, C, K, 3, code) == int(folds)
    rows = L.kp_conv_fwd_stats_rows(N, H + 2, W + 2, C, K, 3, code)
    nrows = (N * H * W + 63) // 64

    def run(fused):
        y = torch.full((N, H, W, K), float('nan'), device='cuda', dtype=dt)
        stats = torch.full((nrows, 2, K), float('nan'), device='cuda')
        mean = torch.full((K,), float('nan'), device='cuda')
        invstd = torch.full((K,), float('nan'), device='cuda')
        rm, rv = torch.linspace(-1, 1, K).cuda(), torch.linspace(0.5, 2, K).cuda()
        nbt = torch.tensor(7, device='cuda')
        ws = torch.empty(128 * K, device='cuda', dtype=torch.float64)
        tickets = torch.zeros(64, device='cuda', dtype=torch.int32)
        if fused:
            call('kp_conv_fwd_fin', ptr(x), ptr(w), ptr(bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, code, ptr(ws), ptr(mean), ptr(invstd),
                 ptr(rm), ptr(rv), ptr(nbt), 0.1, 1e-5, N * H * W, ptr(tickets), st)
        else:
            call('kp_conv_fwd', ptr(x), ptr(w), ptr(bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, code, st)
            assert _lib.last_rows() == rows
            call('kp_bn_finalize', ptr(stats), rows, ptr(ws), ptr(mean), ptr(invstd), ptr(rm), ptr(rv), ptr(nbt), 0.1, 1e-5, N * H * W, K, st)
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0
        return y, stats[:rows].clone(), mean, invstd, rm, rv, int(nbt)

    a, b = run(True), run(False)
    assert torch.equal(a[0].view(torch.int16), b[0].view(torch.int16))
    assert torch.equal(a[1], b[1])
    for i in (2, 3, 4, 5):
        assert torch.isfinite(a[i]).all() and _ulp_close(a[i], b[i], 2), i
    assert a[6] == b[6] == 8
    # ... and against the exact statistics of the stored tensor's fp32 accumulators (fp64 sums of the rows)
    s = a[1].double().sum(0)
    mu = s[0] / (N * H * W)
    var = (s[1] / (N * H * W) - mu * mu).clamp_min(0)
    assert _ulp_close(a[2], mu.float(), 2) and _ulp_close(a[3], (1.0 / (var + 1e-5).sqrt()).float(), 2)
    # deterministic: the same call again gives the same bits
    c = run(True)
    assert all(torch.equal(a[i], c[i]) for i in (2, 3, 4, 5))
    # the A/B knob takes the fold away (and the call still does both jobs)
    call('kp_config', KP_CFG_FIN_FOLD, 0)
    try:
        assert L.kp_conv_fwd_fin_folds(N, H + 2, W + 2, C, K, 3, code) == 0
        d = run(True)
    finally:
        call('kp_config', KP_CFG_FIN_FOLD, 1)
    assert all(torch.equal(b[i], d[i]) for i in (0, 1, 2, 3, 4, 5))


def test_two_folded_launches_on_two_streams_and_many_in_a_row():
    """Tickets are per stream (runtime.ticket_ws): two encoders' convolutions finalize side by side; forty launches in a row on one
    stream reuse one ticket buffer (each leaves it zero)."""
    from keypoints_amd import _lib, runtime
    from keypoints_amd._lib import call, ptr
    code = _lib.KP_F16
    N, H, W, C, K = 16, 64, 64, 128, 256
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(N, H + 2, W + 2, C, generator=g).cuda().half() for _ in range(2)]
    w = (torch.randn(K, 3, 3, C, generator=g) * 0.03).cuda().half()
    bias = torch.zeros(K).cuda()
    nrows = (N * H * W + 63) // 64
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for rep in range(20):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                y = torch.empty(N, H, W, K, device='cuda', dtype=torch.float16)
                stats = torch.empty(nrows, 2, K, device='cuda')
                mean, invstd = torch.empty(K, device='cuda'), torch.empty(K, device='cuda')
                ws = torch.empty(128 * K, device='cuda', dtype=torch.float64)
                tk = runtime.ticket_ws(xs[i].device)
                call('kp_conv_fwd_fin', ptr(xs[i]), ptr(w), ptr(bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, code, ptr(ws), ptr(mean),
                     ptr(invstd), None, None, None, 0.1, 1e-5, N * H * W, ptr(tk), s.cuda_stream)
                outs[i].append((mean, invstd, y, stats, ws))
    torch.cuda.synchronize()
    assert runtime.ticket_ws is not None
    for i in range(2):
        m0, i0 = outs[i][0][0], outs[i][0][1]
        assert torch.isfinite(m0).all() and torch.isfinite(i0).all()
        for mean, invstd, *_ in outs[i][1:]:
            assert torch.equal(mean, m0) and torch.equal(invstd, i0)
    assert not torch.equal(outs[0][0][0], outs[1][0][0])


def test_unit_forward_with_and_without_the_fold_agree():
    """Model F's encoder Unit (forward + backward, train-mode BatchNorm) with runtime.set_fold_finalize on / off: outputs, running
    statistics and gradients agree to the fp32 rounding of the statistics (the same rows, summed in another fixed order)."""
    import keypoints_amd as ka
    from keypoints_amd.models import transporter
    g = torch.Generator().manual_seed(3)
    x = torch.rand(8, 3, 128, 128, generator=g).cuda()
    res = []
    for fold in (True, False):
        ka.set_fold_finalize(fold)
        try:
            torch.manual_seed(7)
            net = transporter.make('F', 3, 64, 10).cuda().set_precision('fp16')
            out = net.feature(x)
            out.square().mean().backward()
            torch.cuda.synchronize()
            res.append((out.detach().clone(), {k: v.detach().clone() for k, v in net.feature.state_dict().items()},
                        [p.grad.detach().clone() for p in net.feature.parameters()]))
        finally:
            ka.set_fold_finalize(True)
    (o0, sd0, g0), (o1, sd1, g1) = res
    assert float((o0 - o1).norm() / o1.norm()) <= 2e-3      # (16-bit storage: a statistic's last bit flips single roundings downstream)
    for k in sd0:
        if 'running' in k:
            assert float((sd0[k] - sd1[k]).abs().max()) <= 1e-5 * max(1.0, float(sd1[k].abs().max())), k
        elif 'num_batches' in k:
            assert int(sd0[k]) == int(sd1[k]) == 1
    for a_, b_ in zip(g0, g1):
        assert float((a_ - b_).norm()) <= 2e-2 * float(b_.norm()) + 1e-8


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
def test_store_stage_statistics_knob(dt):
    """KP_CFG_STATS_STORE = 1 (measurement, not the default): the 128-channel-tile forward kernel takes its BatchNorm sums at the store
    stage from the STORED values, in 16 registers instead of 64 (206 instead of 250: room on a CU for the other encoder chain's pass
    kernels; model F step -0.22 ms, docs/experiments/r06.md section 10).  The convolution output is bit-identical; the sums are the
    exact sums of the rounded tensor -- i.e. they differ from the default's fp32-accumulator sums by the storage rounding only."""
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.runtime import stream_ptr
    KP_CFG_STATS_STORE = 18
    code = _lib.KP_F16 if dt == torch.float16 else _lib.KP_BF16
    N, H, W, C, K = 16, 64, 64, 128, 256
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, H + 2, W + 2, C, generator=g).cuda().to(dt)
    w = (torch.randn(K, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda().to(dt)
    bias = torch.randn(K, generator=g).cuda()
    st = stream_ptr()
    res = []
    for knob in (0, 1):
        call('kp_config', KP_CFG_STATS_STORE, knob)
        try:
            y = torch.full((N, H, W, K), float('nan'), device='cuda', dtype=dt)
            stats = torch.full(((N * H * W + 63) // 64, 2, K), float('nan'), device='cuda')
            call('kp_conv_fwd', ptr(x), ptr(w), ptr(bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, code, st)
            rows = _lib.last_rows()
            torch.cuda.synchronize()
            res.append((y, stats[:rows].double().sum(0)))
        finally:
            call('kp_config', KP_CFG_STATS_STORE, 0)
    (y0, s0), (y1, s1) = res
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    yd = y1.double().reshape(-1, K)
    exact = torch.stack((yd.sum(0), (yd * yd).sum(0)))
    scale = torch.stack((yd.abs().sum(0), (yd * yd).sum(0)))
    assert float(((s1 - exact).abs() / scale).max()) <= 2e-6            # the sums OF the stored tensor, to fp32 summation accuracy
    ulp = 2.0 ** -11 if dt == torch.float16 else 2.0 ** -8
    assert float(((s1 - s0).abs() / scale).max()) <= 0.05 * ulp         # = the default's sums up to the (zero-mean) storage rounding
