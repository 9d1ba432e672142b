"""Does the power-limited MFMA loop of the patch convolution have issue slots for VALU work?  The measurement build of
conv3x3_patch_k (kp_config(KP_CFG_DEBUG, mask != 0)) adds 24 / 48 / 96 / 168 independent fp32 FMAs per (tap, chunk) step and
wave: what a BatchNorm-apply of the LDS patch fused into the loop would issue is ~22 per step.
python tools/probes/valu_in_loop.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_layers import timed
N = 128; st = torch.cuda.current_stream().cuda_stream; bf = torch.bfloat16
for (C, K, H) in [(256, 256, 64), (512, 512, 32), (256, 128, 128)]:
    a = (torch.randn(N, H + 2, H + 2, C, device='cuda') * 0.5).to(bf); wf = (torch.randn(K, 3, 3, C, device='cuda') * 0.05).to(bf)
    bias = torch.zeros(K, device='cuda'); y = torch.empty(N, H, H, K, device='cuda', dtype=bf); stats = torch.empty(N * H * H // 64, 2, K, device='cuda')
    fl = 2.0 * N * H * H * 9 * C * K
    for mask, what in [(0, 'production kernel'), (8, 'measurement build, no extra work'), (8 | 0x100, '+24 FMA / step'), (8 | 0x200, '+48'),
                       (8 | 0x400, '+96'), (8 | 0x700, '+168')]:
        call('kp_config', 1, mask)
        t = timed(lambda: call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3, 1, st), 5)
        print(f'C={C} K={K} H={H} {what:34s}: {t * 1e6:8.1f} us  {fl / t / 1e12:7.0f} TFLOP/s')
    call('kp_config', 1, 0)
