#!/usr/bin/env python3
"""Timings of the 3x3 kernels at shapes NO BASELINE network has (round-4 verdict, missing item 5: "parity-tested, never timed"):
64-multiple channels on maps that are not multiples of 16 (linear implicit-GEMM forward, generic data gradient, per-tap weight
gradient) and the 8-channel MAPPER layers of vgg.py:69 (generic VALU kernels), next to the same channels on a 16-multiple map.
    python tools/probes/offshape_layers.py [--f16]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr

ap = argparse.ArgumentParser(); ap.add_argument('--f16', action='store_true'); ap.add_argument('--batch', type=int, default=128)
args = ap.parse_args()
bf = torch.float16 if args.f16 else torch.bfloat16
DT = _lib.KP_F16 if args.f16 else _lib.KP_BF16
st = torch.cuda.current_stream().cuda_stream
N = args.batch


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


CASES = [('64>128 @128x128 (BASELINE shape)', 64, 128, 128, 128), ('64>128 @100x100', 64, 128, 100, 100), ('64>128 @120x136', 64, 128, 120, 136),
         ('256>256 @64x64 (BASELINE shape)', 256, 256, 64, 64), ('256>256 @40x40', 256, 256, 40, 40), ('256>256 @50x70', 256, 256, 50, 70),
         ('512>512 @16x16 (BASELINE shape)', 512, 512, 16, 16), ('512>512 @12x12', 512, 512, 12, 12),
         ('MAPPER 8>8 @84x84', 8, 8, 84, 84), ('MAPPER 8>8 @16x16', 8, 8, 16, 16), ('Pong 16>32 @84x84 (BASELINE shape)', 16, 32, 84, 84)]
print(f'{"layer (batch %d)" % N:<38}{"fwd":>16}{"dgrad":>16}{"wgrad":>16}   TFLOP/s (algorithmic GB/s)')
for name, C, K, H, W in CASES:
    flops = 2.0 * N * H * W * 9 * C * K
    nbytes = 2.0 * N * (H * W * K + (H + 2) * (W + 2) * C)
    a = (torch.randn(N, H + 2, W + 2, C, device='cuda') * 0.5).to(bf)
    wf = (torch.randn(K, 3, 3, C, device='cuda') * 0.05).to(bf)
    wd = (torch.randn(C, 3, 3, K, device='cuda') * 0.05).to(bf)
    bias = torch.zeros(K, device='cuda')
    y = torch.empty(N, H, W, K, device='cuda', dtype=bf)
    stats = torch.empty((N * H * W + 63) // 64, 2, K, device='cuda')
    dy = torch.zeros(N, H + 4, W + 4, K, device='cuda', dtype=bf)
    dy[:, 2:-2, 2:-2] = (torch.randn(N, H, W, K, device='cuda') * 0.5).to(bf)
    gx = torch.empty(N, H + 2, W + 2, C, device='cuda', dtype=bf)
    parts = _lib.lib().kp_conv_wgrad_partials(N, H, W, C, K, 3, 2, DT)
    scratch = torch.zeros(parts, K, 3, 3, C, device='cuda')
    tf = timed(lambda: call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, DT, st))
    td = timed(lambda: call('kp_conv_dgrad', ptr(dy), ptr(wd), ptr(gx), N, H, W, K, C, DT, st))
    tw = timed(lambda: call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, W, C, K, 3, 2, DT, st))
    cell = lambda t: f'{flops / t / 1e12:7.1f} ({nbytes / t / 1e9:5.0f})'
    print(f'{name:<38}{cell(tf):>16}{cell(td):>16}{cell(tw):>16}   partials {parts}')
