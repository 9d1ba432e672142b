"""BatchNorm-backward passes alone at config-2 shapes: plain (gp 0) and padded-domain (gp 1) gradient: python tools/probes/bn_apply_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
from bench_layers import timed
N, H = 256, 84; bf = torch.bfloat16; st = torch.cuda.current_stream().cuda_stream
for C in (16, 32):
    y = torch.randn(N, H, H, C, device='cuda').to(bf)
    bn = [torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5, torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1]
    sums = torch.randn(2 * C, device='cuda')
    out = []
    for gp in (0, 1):
        g = torch.randn(N, H + 2 * gp, H + 2 * gp, C, device='cuda').to(bf)
        for dp in (0, 2):
            dy = torch.empty(N, H + 2 * dp, H + 2 * dp, C, device='cuda', dtype=bf)
            t = timed(lambda: call('kp_bn_act_bwd_apply', ptr(g), gp, _lib.TAIL_NONE, None, ptr(y), *[ptr(b) for b in bn], 0.01, ptr(sums), ptr(dy), dp, N, H, H, C, 1, st), 10)
            gb = (g.numel() + y.numel() + dy.numel()) * 2 / 1e9
            out.append(f'gp {gp} dp {dp}: {t * 1e6:5.1f} us ({gb / t:4.0f} GB/s)')
        part = torch.empty(N * H, 2, C, device='cuda')
        t = timed(lambda: call('kp_bn_act_bwd_reduce', ptr(g), gp, _lib.TAIL_NONE, None, ptr(y), *[ptr(b) for b in bn], 0.01, ptr(part), N, H, H, C, 1, st), 10)
        out.append(f'gp {gp} reduce: {t * 1e6:5.1f} us ({(g.numel() + y.numel()) * 2 / 1e9 / t:4.0f} GB/s)')
    print(f'C={C}: ' + ' | '.join(out))
