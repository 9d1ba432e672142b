#!/usr/bin/env python3
"""Single-channel first layer (BASELINE config 2: 1 -> 16 filters on 84 x 84 frames with their border, 256 pairs): the row kernel
(KP_CFG_FIRST_ROWS, workgroups per launch) against conv3_first_fwd_k (0), single launch and the two-half launch.
    python tools/probes/first_rows.py [--f16]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr

ap = argparse.ArgumentParser(); ap.add_argument('--f16', action='store_true'); ap.add_argument('--batch', type=int, default=256)
args = ap.parse_args()
dt = torch.float16 if args.f16 else torch.bfloat16
DT = _lib.KP_F16 if args.f16 else _lib.KP_BF16
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for K in (16, 32):
    for Hi in (88,):
        Nh = args.batch
        x = torch.randn(2 * Nh, Hi, Hi, 1, device='cuda').to(dt)
        w = (torch.randn(K, 3, 3, 1, device='cuda') * 0.4).to(dt)
        bias = torch.zeros(K, device='cuda')
        y = torch.empty(2 * Nh, Hi - 2, Hi - 2, K, device='cuda', dtype=dt)
        stats = torch.empty(2 * ((Nh * (Hi - 2) ** 2 + 63) // 64), 2, K, device='cuda')
        mb = 2.0 * Nh * ((Hi - 2) ** 2 * K + Hi * Hi) / 1e6
        for knob in (0, 128, 256, 384, 512, 768, 1024, 2048):
            call('kp_config', 14, knob)
            t1 = timed(lambda: call('kp_conv_fwd', ptr(x), ptr(w), ptr(bias), ptr(y), ptr(stats), Nh, Hi, Hi, 1, K, 3, DT, st))
            r1 = _lib.last_rows()
            t2 = timed(lambda: call('kp_conv_fwd_pair', ptr(x), ptr(w), ptr(bias), ptr(y), ptr(stats), 2 * Nh, Hi, Hi, 1, K, DT, st))
            print(f'K={K} {Hi - 4}x{Hi - 4} batch {Nh}  knob {knob:5d}: single {t1:6.1f} us ({mb / t1:5.2f} TB/s, rows {r1})   '
                  f'two halves {t2:6.1f} us ({2 * mb / t2:5.2f} TB/s)')
call('kp_config', 14, 1)
