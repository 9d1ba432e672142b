#!/usr/bin/env python3
"""Per layer at the bench batch: data gradient + separate BatchNorm-backward reduce pass  vs  kp_conv_dgrad_bnred (the sums in the
data gradient's epilogue), for model F's layers whose producer has no pool / upsample tail.  us per launch, fp16."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

N = 128
L = _lib.lib()
code = _lib.KP_F16
st = torch.cuda.current_stream().cuda_stream
# (name, H, K = gradient channels of layer L, C = its input channels = channels of the layer below, times per step)
LAYERS = [('enc 64>128 @128', 128, 128, 64, 2), ('enc 256>256 @64', 64, 256, 256, 2), ('enc 512>512 @32', 32, 512, 512, 2), ('enc 512>512 @16', 16, 512, 512, 2),
          ('dec 512>512 @16', 16, 512, 512, 1), ('dec 256>256 @32', 32, 256, 256, 1), ('dec 256>256 @64', 64, 256, 256, 1), ('dec 128>64 @128', 128, 64, 128, 1)]
tot = [0.0, 0.0, 0.0]
for name, H, K, C, times in LAYERS:
    W = H
    dy = torch.zeros(N, H + 4, W + 4, K, device='cuda', dtype=torch.float16)
    dy[:, 2:-2, 2:-2] = (torch.randn(N, H, W, K, device='cuda') * 0.1).half()
    wd = (torch.randn(C, 3, 3, K, device='cuda') * 0.05).half()
    yred = torch.randn(N, H, W, C, device='cuda').half()
    bn = [(torch.randn(C, device='cuda') * 0.3), (torch.rand(C, device='cuda') + 0.5), (torch.randn(C, device='cuda') * 0.5 + 1), (torch.randn(C, device='cuda') * 0.3)]
    gx = torch.empty(N, H + 2, W + 2, C, device='cuda', dtype=torch.float16)
    p0 = torch.empty(N * H, 2, C, device='cuda')
    r1 = L.kp_conv_dgrad_bnred_rows(N, H, W, K, C, code)
    p1 = torch.empty(max(r1, 1), 2, C, device='cuda')
    td = timed(lambda: call('kp_conv_dgrad', ptr(dy), ptr(wd), ptr(gx), N, H, W, K, C, code, st))
    tr = timed(lambda: call('kp_bn_act_bwd_reduce', ptr(gx), 1, _lib.TAIL_NONE, None, ptr(yred), *[ptr(t) for t in bn], 0.01, ptr(p0), N, H, W, C, code, st))
    tf = timed(lambda: call('kp_conv_dgrad_bnred', ptr(dy), ptr(wd), ptr(gx), ptr(yred), *[ptr(t) for t in bn], 0.01, ptr(p1), N, H, W, K, C, code, st)) if r1 > 0 else float('nan')
    print(f'{name:<18} dgrad {td:7.1f} us + reduce {tr:6.1f} us = {td + tr:7.1f}   fused {tf:7.1f}   ({tf - td - tr:+6.1f} us, x{times} per step)')
    tot[0] += times * td; tot[1] += times * tr; tot[2] += times * tf
print(f'per step: dgrad {tot[0] / 1e3:.2f} ms + reduce {tot[1] / 1e3:.2f} ms = {(tot[0] + tot[1]) / 1e3:.2f} ms; fused {tot[2] / 1e3:.2f} ms')
