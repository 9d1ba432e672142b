"""Per-kernel times of the 16/32-channel (Pong-class) kernels at config-2 shapes (256 x 84 x 84): python tools/probes/small_time.py [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypoints_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from keypoints_amd._lib import call, ptr
from bench_layers import timed
N, H = 256, 84; bf = torch.bfloat16; st = torch.cuda.current_stream().cuda_stream
out = []
for C, K in ((16, 16), (16, 32), (32, 16), (32, 32)):
    a = (torch.randn(N, H + 2, H + 2, C, device='cuda') * 0.5).to(bf); wf = (torch.randn(K, 3, 3, C, device='cuda') * 0.1).to(bf)
    bias = torch.zeros(K, device='cuda'); y = torch.empty(N, H, H, K, device='cuda', dtype=bf); stats = torch.empty(N * H * H // 64 + 1, 2, K, device='cuda')
    dy = torch.zeros(N, H + 4, H + 4, K, device='cuda', dtype=bf); dy[:, 2:-2, 2:-2] = torch.randn(N, H, H, K, device='cuda').to(bf)
    parts = _lib.lib().kp_conv_wgrad_partials(N, H, H, C, K, 3, 2, 1)
    scratch = torch.empty(parts, K, 3, 3, C, device='cuda')
    tf = timed(lambda: call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3, 1, st), 10)
    tw = timed(lambda: call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 2, 1, st), 10)
    yp = (torch.randn(N, H, H, C, device='cuda') * 0.5).to(bf)
    bnp = [torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5, torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1]
    tfb = timed(lambda: call('kp_conv_fwd_bnin', ptr(yp), *[ptr(t) for t in bnp], 0.01, ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H, H, C, K, 1, st), 10)
    twb = timed(lambda: call('kp_conv_wgrad_bnin', ptr(yp), *[ptr(t) for t in bnp], 0.01, ptr(dy), ptr(scratch), N, H, H, C, K, 2, 1, st), 10)
    gx = torch.empty(N, H + 2, H + 2, C, device='cuda', dtype=bf); wd = (torch.randn(C, 3, 3, K, device='cuda') * 0.1).to(bf)
    td = timed(lambda: call('kp_conv_dgrad', ptr(dy), ptr(wd), ptr(gx), N, H, H, K, C, 1, st), 10)
    gb = (a.numel() + y.numel()) * 2 / 1e9
    out.append(f'{C}->{K}: fwd {tf * 1e6:5.1f} ({gb / tf:4.0f} GB/s) fwd_bnin {tfb * 1e6:5.1f} dgrad(K->C) {td * 1e6:5.1f} wgrad {tw * 1e6:5.1f} wgrad_bnin {twb * 1e6:5.1f}')
print(' | '.join(out))
