"""First-layer forward (1 -> 16 at config-2 shape): with / without the statistics rows: python tools/probes/first_time2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
from bench_layers import timed
N, H = 256, 84; bf = torch.bfloat16; st = torch.cuda.current_stream().cuda_stream
for C, K in ((1, 16), (1, 32)):
    a = (torch.randn(N, H + 2, H + 2, C, device='cuda') * 0.5).to(bf); wf = (torch.randn(K, 3, 3, C, device='cuda') * 0.1).to(bf)
    bias = torch.zeros(K, device='cuda'); y = torch.empty(N, H, H, K, device='cuda', dtype=bf); stats = torch.empty(N * H * H // 64 + 1, 2, K, device='cuda')
    t1 = timed(lambda: call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3, 1, st), 10)
    t0 = timed(lambda: call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), None, N, H + 2, H + 2, C, K, 3, 1, st), 10)
    print(f'{C}->{K}: fwd with statistics {t1 * 1e6:5.1f} us, without {t0 * 1e6:5.1f} us; output {N * H * H * K * 2 / 1e6:.0f} MB')
