#!/usr/bin/env python3
"""Energy per FLOP of the three forward / data-gradient kernel generations on ONE layer shape, run back to back for ~2 s each
with the socket power and shader clock sampled every 10 ms (bench.py's PowerSampler: amdgpu hwmon):
    patch (first generation, the default)    pair (128 x 64 wave tile)    wino (Winograd F(2,3), 1.5x fewer MFMAs)
-> TFLOP/s (direct-equivalent), W, GHz, J per direct-equivalent TFLOP.  Says whether a variant is slower because it draws the same
power for less work (energy-limited) or because it cannot use the power (issue / LDS-limited: lower W or higher clock)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
from bench import PowerSampler

N = 128
DT, tdt = _lib.KP_F16, torch.float16
st = torch.cuda.current_stream().cuda_stream
L = _lib.lib()
for name, C, K, H in [('256>256 @64', 256, 256, 64), ('512>512 @32', 512, 512, 32), ('64>128 @128', 64, 128, 128)]:
    W = H
    flops = 2.0 * N * H * W * 9 * C * K
    a = (torch.randn(N, H + 2, W + 2, C, device='cuda') * 0.5).to(tdt)
    wf = (torch.randn(K, 3, 3, C, device='cuda') * 0.05).to(tdt)
    bias = torch.zeros(K, device='cuda')
    y = torch.empty(N, H, W, K, device='cuda', dtype=tdt)
    stats = torch.empty(max(N * H * W // 64, 4096), 2, K, device='cuda')
    uf = torch.empty(K, 4, 3, C, device='cuda', dtype=tdt)
    call('kp_conv_wino_pack', ptr(wf), ptr(uf), K, C, DT, st)
    zeros = torch.zeros_like(a)
    variants = [('patch', lambda x: (call('kp_config', 0, 1), call('kp_conv_fwd', ptr(x), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, DT, st))),
                ('pair', lambda x: (call('kp_config', 0, 2), call('kp_conv_fwd', ptr(x), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, DT, st))),
                ('wino', lambda x: call('kp_conv_fwd_wino', ptr(x), ptr(uf), ptr(bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, DT, st)),
                ('patch, zero input', None)]
    for vname, fn in variants:
        x = a
        if fn is None:
            fn, x = variants[0][1], zeros
        for _ in range(20):
            fn(x)
        torch.cuda.synchronize()
        ps = PowerSampler(torch.device('cuda', 0))
        ps.start()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.0:
            for _ in range(50):
                fn(x)
            torch.cuda.synchronize()
            n += 50
        dt = time.perf_counter() - t0
        p = ps.stop() or {}
        tf = flops * n / dt / 1e12
        w, clk = p.get('avg_socket_w', float('nan')), p.get('avg_sclk_mhz', float('nan'))
        print(f'{name:<14} {vname:<18} {tf:7.0f} TFLOP/s (direct-equivalent)  {w:6.0f} W  {clk / 1e3:5.2f} GHz  {w / tf:6.3f} J/TFLOP  ({1e6 * dt / n:6.1f} us per launch)')
    call('kp_config', 0, 1)
