#!/usr/bin/env python3
"""Per-layer timing of the 3x3 convolution kernels at the shapes of model F (bench workload: 128 pairs,
128x128): forward, data-gradient (padded-domain output) and weight-gradient, through the C ABI.

    python tools/bench_layers.py [--batch 128] [--reps 5] [--ab]     (--ab: also time KP_CFG_CONV_PATCH=0)

Prints one line per layer with TFLOP/s per kernel, and the FLOP-weighted totals.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr

# (name, C, K, H) distinct 3x3 layers of Transporter F at 128x128 input, with how many times each runs per
# train step as (fwd, dgrad, wgrad): encoder-type Units run 4x forward (2 nets x source/target) and 2x backward.
LAYERS = [
    ('enc 64>128 @128', 64, 128, 128, 4, 2, 2), ('enc 128>256 @64', 128, 256, 64, 4, 2, 2),
    ('enc 256>256 @64', 256, 256, 64, 4, 2, 2), ('enc 256>512 @32', 256, 512, 32, 4, 2, 2),
    ('enc 512>512 @32', 512, 512, 32, 4, 2, 2), ('enc 512>512 @16', 512, 512, 16, 8, 4, 4),
    ('dec 64>512 @16', 64, 512, 16, 1, 0, 1), ('dec 512>512 @16', 512, 512, 16, 1, 1, 1),
    ('dec 512>256 @32', 512, 256, 32, 1, 1, 1), ('dec 256>256 @32', 256, 256, 32, 1, 1, 1),
    ('dec 256>256 @64', 256, 256, 64, 2, 2, 2), ('dec 256>128 @128', 256, 128, 128, 1, 1, 1),
    ('dec 128>64 @128', 128, 64, 128, 1, 1, 1),
]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--ab', action='store_true')
    ap.add_argument('--only', default='')
    ap.add_argument('--modes', default='', help='comma list of KP_CFG_CONV_PATCH values to time side by side (2 = pair kernel, 1 = first-generation patch kernel, 0 = linear)')
    ap.add_argument('--f16', action='store_true', help='fp16 storage instead of bf16')
    ap.add_argument('--lib', default='', help='another build of libkp_hip.so to time (A/B between builds)')
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    N = args.batch
    st = torch.cuda.current_stream().cuda_stream
    bf = torch.float16 if args.f16 else torch.bfloat16
    DT = _lib.KP_F16 if args.f16 else _lib.KP_BF16
    tot = {}
    modes = [int(m) for m in args.modes.split(',')] if args.modes else ([1, 0] if args.ab else [1])
    print(f'{"layer":<20}' + ''.join(f'{"fwd":>9}{"dgrad":>9}' for _ in modes) + f'{"wgrad":>9}   (TFLOP/s)')
    for name, C, K, H, nf, nd, nw in LAYERS:
        if args.only and args.only not in name:
            continue
        W = H
        flops = 2.0 * N * H * W * 9 * C * K
        a = (torch.randn(N, H + 2, W + 2, C, device='cuda') * 0.5).to(bf)
        wf = (torch.randn(K, 3, 3, C, device='cuda') * 0.05).to(bf)
        wd = (torch.randn(C, 3, 3, K, device='cuda') * 0.05).to(bf)
        bias = torch.zeros(K, device='cuda')
        y = torch.empty(N, H, W, K, device='cuda', dtype=bf)
        stats = torch.empty(N * H * W // 64, 2, K, device='cuda')
        dy = torch.zeros(N, H + 4, W + 4, K, device='cuda', dtype=bf)
        dy[:, 2:-2, 2:-2] = (torch.randn(N, H, W, K, device='cuda') * 0.5).to(bf)
        gx = torch.empty(N, H + 2, W + 2, C, device='cuda', dtype=bf)
        scratch = torch.zeros(_lib.lib().kp_conv_wgrad_partials(N, H, W, C, K, 3, 2, DT), K, 3, 3, C, device='cuda')
        cols = []
        for mode in modes:
            if True:
                call('kp_config', 0, mode)
                tf = timed(lambda: call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K,
                                        3, DT, st), args.reps)
                td = timed(lambda: call('kp_conv_dgrad', ptr(dy), ptr(wd), ptr(gx), N, H, W, K, C, DT, st), args.reps)
            cols += [flops / tf / 1e12, flops / td / 1e12]   # dgrad priced at its algorithmic (interior) FLOPs
            key = f'patch={mode}'
            t = tot.setdefault(key, [0.0, 0.0])
            t[0] += nf * flops + nd * flops   # algorithmic (interior) flops
            t[1] += nf * tf + nd * td
        call('kp_config', 0, 1)
        tw = timed(lambda: call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, W, C, K, 3, 2, DT, st),
                   args.reps)
        t = tot.setdefault('wgrad', [0.0, 0.0])
        t[0] += nw * flops
        t[1] += nw * tw
        print(f'{name:<20}' + ''.join(f'{c:9.0f}' for c in cols) + f'{flops / tw / 1e12:9.0f}')
        del a, wf, wd, y, stats, dy, gx, scratch
    if not args.only or 'first' in args.only:
        C, K, H = 3, 64, 128
        a = (torch.randn(N, H + 2, H + 2, C, device='cuda') * 0.5).to(bf)
        wf = (torch.randn(K, 3, 3, C, device='cuda') * 0.2).to(bf)
        bias = torch.zeros(K, device='cuda')
        y = torch.empty(N, H, H, K, device='cuda', dtype=bf)
        stats = torch.empty(N * H * H // 64, 2, K, device='cuda')
        dy = (torch.randn(N, H, H, K, device='cuda') * 0.5).to(bf)
        scratch = torch.zeros(_lib.lib().kp_conv_wgrad_partials(N, H, H, C, K, 3, 0, DT), K, 3, 3, C, device='cuda')
        tf = timed(lambda: call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3,
                                DT, st), args.reps)
        tw = timed(lambda: call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 0, DT, st),
                   args.reps)
        gb = N * H * H * K * 2 / 1e9
        print(f'first 3>64 @128: fwd {tf * 1e6:7.0f} us ({gb / tf:6.0f} GB/s written)   wgrad {tw * 1e6:7.0f} us '
              f'({gb / tw:6.0f} GB/s read)   x4 fwd + x2 wgrad per step = {(4 * tf + 2 * tw) * 1e3:.2f} ms')
    for k, (f, t) in tot.items():
        print(f'total {k}: {t * 1e3:8.2f} ms/step-equivalent   {f / t / 1e12:7.0f} TFLOP/s')


if __name__ == '__main__':
    main()
