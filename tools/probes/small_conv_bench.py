#!/usr/bin/env python3
"""Times kp_conv_fwd / kp_conv_wgrad at the Pong-class layer shapes (C, K in {16, 32}, 256 x 84 x 84) and checks the forward
against torch.  KP_CONV_SMALL_LDS=0 selects the register-only kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
from keypoints_amd.runtime import stream_ptr

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

N, H = 256, 84
call('kp_config', 1, int(os.environ.get('DBG', 0)))
for C, K in [(16, 32), (32, 32), (32, 16), (16, 16)]:
    g = torch.Generator(device='cuda').manual_seed(C + K)
    x = torch.randn(N, H + 2, H + 2, C, device='cuda', generator=g).bfloat16()
    w = (torch.randn(K, 3, 3, C, device='cuda', generator=g) * (2.0 / (9 * C)) ** 0.5).bfloat16()
    b = torch.randn(K, device='cuda', generator=g)
    y = torch.full((N, H, H, K), float('nan'), device='cuda', dtype=torch.bfloat16)
    stats = torch.full(((N * H * H + 63) // 64, 2, K), float('nan'), device='cuda')
    st = stream_ptr()
    f = lambda: call('kp_conv_fwd', ptr(x), ptr(w), ptr(b), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3, 1, st)
    t = timeit(f)
    rows = _lib.last_rows()
    ref = F.conv2d(x[:8].float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b).permute(0, 2, 3, 1)
    err = (y[:8].float() - ref).abs()
    ok = bool(torch.isfinite(y.float()).all()) and float((err - (2 ** -7 * ref.abs() + 2e-4)).max()) <= 0
    full = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b)
    s = stats[:rows].double().sum(0)
    sok = float(((s[0] - full.double().sum((0, 2, 3))).abs() / full.double().abs().sum((0, 2, 3))).max())
    by = N * ((H + 2) ** 2 * C + H * H * K) * 2
    dy = torch.zeros(N, H + 4, H + 4, K, device='cuda', dtype=torch.bfloat16)
    dy[:, 2:-2, 2:-2] = torch.randn(N, H, H, K, device='cuda', generator=g).bfloat16()
    parts = _lib.lib().kp_conv_wgrad_partials(N, H, H, C, K, 3, 2, 1)
    scratch = torch.zeros(parts, K, 3, 3, C, device='cuda')
    tw = timeit(lambda: call('kp_conv_wgrad', ptr(x), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 2, 1, st))
    print(f'C={C} K={K}: fwd {t:.0f} us ({by / t / 1e6:.2f} TB/s) rows={rows} ok={ok} stats_err={sok:.1e}   wgrad {tw:.0f} us', flush=True)
