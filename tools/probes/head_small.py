#!/usr/bin/env python3
"""The K <= 4 image heads (decoder out_block, knn.py:123-125) on their BatchNorm-input forms: forward, input gradient, weight gradient at
model F's decoder head (64 -> 3 on 128 x 128 x 128 images) and config 2's (16 -> 1 on 256 x 84 x 84).
    python tools/probes/head_small.py [--f16] [--lib other/libkp_hip.so]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser(); ap.add_argument('--f16', action='store_true'); ap.add_argument('--lib', default='')
args = ap.parse_args()
import torch
from keypoints_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from keypoints_amd._lib import call, ptr
dt = torch.float16 if args.f16 else torch.bfloat16
DT = _lib.KP_F16 if args.f16 else _lib.KP_BF16
st = torch.cuda.current_stream().cuda_stream


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (N, H, C, K) in [(128, 128, 64, 3), (256, 84, 16, 1), (512, 84, 16, 1), (256, 84, 32, 4)]:
    y = torch.randn(N, H, H, C, device='cuda').to(dt)
    mean, invstd = torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5
    gamma, beta = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1
    w = torch.randn(K, C, device='cuda') * C ** -0.5
    b = torch.randn(K, device='cuda')
    out = torch.empty(N, K, H, H, device='cuda')
    g = torch.randn(N, K, H, H, device='cuda')
    dx = torch.empty_like(y); dw = torch.zeros(K, C, device='cuda'); db = torch.zeros(K, device='cuda')
    ws = torch.empty(_lib.lib().kp_head_bwd_ws_floats(N, H, H, C, K, DT), device='cuda')
    bn = (ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), 0.0)
    tf = timeit(lambda: call('kp_head_fwd_bnin', ptr(y), *bn, ptr(w), ptr(b), 0.0, ptr(out), N, H, H, C, K, DT, st))
    tp = timeit(lambda: call('kp_head_fwd', ptr(y), ptr(w), ptr(b), 0.0, ptr(out), N, H, H, C, K, DT, st))
    tdx = timeit(lambda: call('kp_head_bwd_bnin', ptr(g), ptr(out), ptr(y), *bn, ptr(w), 0.0, ptr(dx), None, None, None, N, H, H, C, K, DT, st))
    tdw = timeit(lambda: call('kp_head_bwd_bnin', ptr(g), ptr(out), ptr(y), *bn, ptr(w), 0.0, None, ptr(dw), ptr(db), ptr(ws), N, H, H, C, K, DT, st))
    M = N * H * H
    bx, bo = M * C * 2, M * K * 4
    print(f'N={N} {H}x{H} C={C} K={K}: fwd(bnin) {tf:6.1f} us ({(bx + bo) / tf / 1e6:.2f} TB/s)  fwd {tp:6.1f}  dx {tdx:6.1f} us ({(bx + 2 * bo) / tdx / 1e6:.2f} TB/s)  '
          f'dw(bnin) {tdw:6.1f} us ({(bx + 2 * bo) / tdw / 1e6:.2f} TB/s)', flush=True)
