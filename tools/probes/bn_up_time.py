#!/usr/bin/env python3
"""kp_bn_act_fwd with the bilinear x2 tail (vgg.py:26, the decoder's 'U' entries) at model F's three shapes, 128 images.
    python tools/probes/bn_up_time.py [--f16] [--lib other/libkp_hip.so]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser(); ap.add_argument('--f16', action='store_true'); ap.add_argument('--lib', default=''); ap.add_argument('--walk', default='')
args = ap.parse_args()
import torch
from keypoints_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from keypoints_amd._lib import call, ptr
dt = torch.float16 if args.f16 else torch.bfloat16
DT = _lib.KP_F16 if args.f16 else _lib.KP_BF16
st = torch.cuda.current_stream().cuda_stream
if args.walk:
    call('kp_config', 5, int(args.walk, 0))


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


N = 128
for H, C in [(16, 512), (32, 256), (64, 256), (64, 128)]:   # model F's three 'U' entries, + the 128-channel shape of profiles/r05n_bn_up.txt
    y = torch.randn(N, H, H, C, device='cuda').to(dt)
    mean, invstd = torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5
    gamma, beta = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1
    a = torch.empty(N, 2 * H + 2, 2 * H + 2, C, device='cuda', dtype=dt)
    t = timeit(lambda: call('kp_bn_act_fwd', ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), 0.0, _lib.TAIL_UPSAMPLE2, ptr(a), 1, None, N, H, H, C, DT, st))
    mb = (y.numel() + a.numel()) * 2 / 1e6
    print(f'{H}x{H} -> {2 * H}x{2 * H}, {C} channels: {t:6.1f} us  {mb / t:5.2f} TB/s   checksum {float(a.float().sum()):.6e}', flush=True)
