#!/usr/bin/env python3
"""HBM throughput of the BatchNorm/activation kernels at the layer shapes of model F (128 pairs): algorithmic bytes
(each operand / result once) divided by the measured time.  python tools/bench_bn.py [--batch 128]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from bench_layers import timed

SHAPES = [(64, 128, 'none'), (128, 128, 'pool'), (256, 64, 'none'), (256, 64, 'pool'), (512, 32, 'none'), (512, 32, 'pool'),
          (512, 16, 'none'), (256, 32, 'up'), (256, 64, 'up'), (128, 128, 'none')]


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--cfg', default='', help='kp_config switches key=value[,key=value]')
    ap.add_argument('--lib', default='', help='another build of libkp_hip.so (A/B between builds)'); args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    for kv in filter(None, args.cfg.split(',')):   # e.g. --cfg 5=0: kp_config(KP_CFG_BN_FLAT, 0)
        call('kp_config', *[int(v) for v in kv.split('=')])
    N = args.batch; st = torch.cuda.current_stream().cuda_stream; bf = torch.bfloat16
    print(f'{"C":>4} {"H":>4} {"tail":>5} | {"fwd us":>8} {"GB/s":>6} | {"reduce us":>9} {"GB/s":>6} | {"apply us":>8} {"GB/s":>6}')
    for C, H, tail in SHAPES:
        tcode = {'none': 0, 'pool': 1, 'up': 2}[tail]
        Ho = H // 2 if tail == 'pool' else 2 * H if tail == 'up' else H
        y = torch.randn(N, H, H, C, device='cuda').to(bf)
        out = torch.empty(N, Ho + 2, Ho + 2, C, device='cuda', dtype=bf)
        am = torch.zeros(N, Ho, Ho, C, device='cuda', dtype=torch.uint8) if tail == 'pool' else None
        mean = torch.zeros(C, device='cuda'); inv = torch.ones(C, device='cuda'); ga = torch.ones(C, device='cuda'); be = torch.zeros(C, device='cuda')
        tf = timed(lambda: call('kp_bn_act_fwd', ptr(y), ptr(mean), ptr(inv), ptr(ga), ptr(be), 0.01, tcode, ptr(out), 1, ptr(am),
                                N, H, H, C, _lib.KP_BF16, st), 10)
        bf_bytes = y.numel() * 2 + out.numel() * 2 + (am.numel() if am is not None else 0)
        line = f'{C:4d} {H:4d} {tail:>5} | {tf*1e6:8.1f} {bf_bytes/tf/1e9:6.0f} |'
        if tail != 'up':
            Hs = Ho
            g = torch.randn(N, Hs + 2, Hs + 2, C, device='cuda').to(bf)
            partials = torch.empty(N * H, 2, C, device='cuda'); sums = torch.zeros(2 * C, device='cuda')
            dy = torch.empty(N, H + 4, H + 4, C, device='cuda', dtype=bf)
            tr = timed(lambda: call('kp_bn_act_bwd_reduce', ptr(g), 1, tcode, ptr(am), ptr(y), ptr(mean), ptr(inv), ptr(ga), ptr(be),
                                    0.01, ptr(partials), N, H, H, C, _lib.KP_BF16, st), 10)
            ta = timed(lambda: call('kp_bn_act_bwd_apply', ptr(g), 1, tcode, ptr(am), ptr(y), ptr(mean), ptr(inv), ptr(ga), ptr(be),
                                    0.01, ptr(sums), ptr(dy), 2, N, H, H, C, _lib.KP_BF16, st), 10)
            rb = g.numel() * 2 + y.numel() * 2 + (am.numel() if am is not None else 0)
            ab = rb + dy.numel() * 2
            line += f' {tr*1e6:9.1f} {rb/tr/1e9:6.0f} | {ta*1e6:8.1f} {ab/ta/1e9:6.0f}'
        print(line)


if __name__ == '__main__':
    main()
