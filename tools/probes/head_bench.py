#!/usr/bin/env python3
"""Times kp_head_fwd / kp_head_bwd (dx, dw separately) at the BASELINE head shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
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

for (N, H, C, K) in [(256, 84, 32, 16), (256, 84, 32, 4), (256, 84, 16, 1), (128, 128, 64, 3), (128, 16, 512, 64), (128, 16, 512, 10)]:
    x = torch.randn(N, H, H, C, device='cuda').bfloat16()
    w = torch.randn(K, C, device='cuda') * C ** -0.5
    b = torch.randn(K, device='cuda')
    out = torch.empty(N, K, H, H, device='cuda')
    g = torch.randn(N, K, H, H, device='cuda')
    dx = torch.empty_like(x); dw = torch.zeros(K, C, device='cuda'); db = torch.zeros(K, device='cuda')
    st = stream_ptr()
    tf = timeit(lambda: call('kp_head_fwd', ptr(x), ptr(w), ptr(b), 0.01, ptr(out), N, H, H, C, K, 1, st))
    tdx = timeit(lambda: call('kp_head_bwd', ptr(g), ptr(out), ptr(x), ptr(w), 0.01, ptr(dx), None, None, None, N, H, H, C, K, 1, st))
    tdw = timeit(lambda: call('kp_head_bwd', ptr(g), ptr(out), ptr(x), ptr(w), 0.01, None, ptr(dw), ptr(db), None, N, H, H, C, K, 1, st))
    ws = torch.empty(_lib.lib().kp_head_bwd_ws_floats(N, H, H, C, K, 1), device='cuda')
    tdet = timeit(lambda: call('kp_head_bwd', ptr(g), ptr(out), ptr(x), ptr(w), 0.01, None, ptr(dw), ptr(db), ptr(ws), N, H, H, C, K, 1, st))
    d1 = torch.zeros(K, C, device='cuda'); d2 = torch.zeros(K, C, device='cuda'); b1 = torch.zeros(K, device='cuda'); b2 = torch.zeros(K, device='cuda')
    call('kp_head_bwd', ptr(g), ptr(out), ptr(x), ptr(w), 0.01, None, ptr(d1), ptr(b1), ptr(ws), N, H, H, C, K, 1, st)
    call('kp_head_bwd', ptr(g), ptr(out), ptr(x), ptr(w), 0.01, None, ptr(d2), ptr(b2), ptr(ws), N, H, H, C, K, 1, st)
    d0 = torch.zeros(K, C, device='cuda'); call('kp_head_bwd', ptr(g), ptr(out), ptr(x), ptr(w), 0.01, None, ptr(d0), None, None, N, H, H, C, K, 1, st)
    print(f'   deterministic dw {tdet:.0f} us, ws {ws.numel()*4/1e6:.1f} MB, bitwise repeat {torch.equal(d1, d2) and torch.equal(b1, b2)}, vs atomics rel {float((d1-d0).norm()/d0.norm()):.1e}')
    M = N * H * H
    bx, bo = M * C * 2, M * K * 4
    print(f'N={N} H={H} C={C} K={K}: fwd {tf:.0f} us ({(bx+bo)/tf/1e6:.2f} TB/s)  dx {tdx:.0f} us ({(bx+2*bo)/tdx/1e6:.2f} TB/s)  dw {tdw:.0f} us ({(bx+2*bo)/tdw/1e6:.2f} TB/s)', flush=True)
