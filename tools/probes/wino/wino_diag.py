#!/usr/bin/env python3
"""Where does kp_conv_fwd_wino differ from its torch restatement?  Error map by output row / column / channel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
from test_gpu_conv_wino import wino_pack_ref, wino_ref

def run(N, H, W, C, K, mode):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H + 2, W + 2, C, generator=g).half()
    w = (torch.randn(K, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).half()
    if mode == 'center':      # only the centre tap: y = x[+1,+1] . w11
        w[:, 0] = 0; w[:, 2] = 0; w[:, :, 0] = 0; w[:, :, 2] = 0
    bias = torch.zeros(K)
    st = torch.cuda.current_stream().cuda_stream
    code = _lib.KP_F16
    call('kp_config', 6, 8)
    u = torch.empty(K, 4, 3, C, device='cuda', dtype=torch.float16)
    wd = w.cuda()
    call('kp_conv_wino_pack', ptr(wd), ptr(u), K, C, code, st)
    y = torch.full((N, H, W, K), float('nan'), device='cuda', dtype=torch.float16)
    call('kp_conv_fwd_wino', ptr(x.cuda()), ptr(u), ptr(bias.cuda()), ptr(y), None, N, H + 2, W + 2, C, K, code, st)
    got = y.float().cpu()
    emu = wino_ref(x, wino_pack_ref(w), bias)
    err = (got - emu).abs()
    err[~torch.isfinite(got)] = 1e9
    bad = err > 1e-2 * emu.abs().max()
    print(f'--- N={N} H={H} W={W} C={C} K={K} {mode}: bad {int(bad.sum())} of {bad.numel()}, nan {int((~torch.isfinite(got)).sum())}')
    print(' by image:', bad.flatten(1).float().mean(1).tolist())
    print(' by row  :', [round(v, 2) for v in bad.float().mean((0, 2, 3)).tolist()])
    print(' by col  :', [round(v, 2) for v in bad.float().mean((0, 1, 3)).tolist()])
    print(' by k/8  :', [round(v, 2) for v in bad.float().mean((0, 1, 2)).reshape(-1, 8).mean(1).tolist()])
    if bad.any():
        i = bad.nonzero()[0].tolist()
        print(' first bad', i, 'got', float(got[tuple(i)]), 'want', float(emu[tuple(i)]))

run(2, 16, 16, 64, 64, 'rand')
run(2, 16, 16, 64, 64, 'center')
run(2, 16, 16, 128, 64, 'rand')
run(2, 16, 16, 64, 128, 'rand')
