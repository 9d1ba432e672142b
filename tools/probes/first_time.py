"""Per-kernel times of the first-layer (Cin = 1 -> 16) kernels at config-2 shape: python tools/probes/first_time.py [lib.so]"""
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
for C, K in ((1, 16), (3, 16), (1, 32)):
    a = (torch.randn(N, H + 2, H + 2, C, device='cuda') * 0.5).to(bf); wf = (torch.randn(K, 3, 3, C, device='cuda') * 0.1).to(bf)
    bias = torch.zeros(K, device='cuda'); y = torch.empty(N, H, H, K, device='cuda', dtype=bf); stats = torch.empty(N * H * H // 64 + 1, 2, K, device='cuda')
    dy = torch.randn(N, H, H, K, device='cuda').to(bf)
    parts = _lib.lib().kp_conv_wgrad_partials(N, H, H, C, K, 3, 0, 1)
    scratch = torch.empty(parts, K, 3, 3, C, device='cuda')
    tf = timed(lambda: call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3, 1, st), 10)
    tw = timed(lambda: call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 0, 1, st), 10)
    print(f'{C}->{K}: fwd {tf * 1e6:5.1f} us  wgrad {tw * 1e6:5.1f} us ({parts} partials)')
