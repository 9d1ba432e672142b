import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
from test_gpu_first_fused import _setup, _finalize
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
L = _lib.lib()
N, C, H, W, dt = 3, 3, 128, 128, 'bf16'
s = _setup(N, C, H, W, dt, 1)
code, st, K = s['code'], s['st'], 64
rows1 = L.kp_conv_first_fused_rows(N, H, W, C, K, code)
print('rows', rows1)
stats1 = torch.full((rows1, 2, K), float('nan'), device='cuda')
call('kp_conv_first_stats', ptr(s['a']), ptr(s['wf']), ptr(s['bias']), ptr(stats1), N, H, W, C, K, code, st)
torch.cuda.synchronize()
nan = torch.isnan(stats1)
print('nan count', int(nan.sum()), 'of', stats1.numel(), 'rows with nan', int(nan.any(dim=2).any(dim=1).sum()))
print('nan by s', nan.sum(dim=(0, 2)).tolist())
print('nan by k (first row)', nan[0].int().tolist())
print('row0 sums', stats1[0, 0, :8].tolist())
y = torch.empty(N, H, W, K, device='cuda', dtype=s['tdt'])
stats0 = torch.zeros((N * H * W + 63) // 64, 2, K, device='cuda')
call('kp_conv_fwd', ptr(s['a']), ptr(s['wf']), ptr(s['bias']), ptr(y), ptr(stats0), N, H + 2, W + 2, C, K, 3, code, st)
rows0 = L.kp_conv_fwd_stats_rows(N, H + 2, W + 2, C, K, 3, code)
print('plain total', stats0[:rows0, 0].sum(0)[:8].tolist())
print('fused total', torch.nan_to_num(stats1)[:, 0].sum(0)[:8].tolist())
