import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
from keypoints_amd.runtime import stream_ptr
torch.set_printoptions(linewidth=200)
def run(N, Ho, Wo, C, K, dp, x, dyi):
    dy = torch.zeros(N, Ho + 2 * dp, Wo + 2 * dp, K, dtype=torch.bfloat16)
    dy[:, dp:dp + Ho, dp:dp + Wo] = dyi
    scratch = torch.zeros(K, 3, 3, C, device='cuda')
    call('kp_conv_wgrad', ptr(x.cuda()), ptr(dy.cuda()), ptr(scratch), N, Ho, Wo, C, K, 3, dp, _lib.KP_BF16, stream_ptr())
    return scratch.cpu()
N, Ho, Wo, C, K, dp = 2, 8, 8, 128, 128, 2
x = torch.ones(N, Ho + 2, Wo + 2, C).bfloat16(); dyi = torch.ones(N, Ho, Wo, K).bfloat16()
g = run(N, Ho, Wo, C, K, dp, x, dyi)
print('all ones: expect', N * Ho * Wo, 'unique', torch.unique(g))
# x = channel index, dy = ones -> expect dW[k][tap][c] = M * c
x = (torch.arange(C).float()[None, None, None, :] * torch.ones(N, Ho + 2, Wo + 2, 1)).bfloat16()
g = run(N, Ho, Wo, C, K, dp, x, dyi)
print('x=c: row k=0 tap0:', (g[0, 0, 0] / (N * Ho * Wo)).tolist()[:40])
print('   all k equal?', bool((g == g[0:1]).all()))
# dy = k index, x ones -> expect M * k
x = torch.ones(N, Ho + 2, Wo + 2, C).bfloat16()
dyi = (torch.arange(K).float()[None, None, None, :] * torch.ones(N, Ho, Wo, 1)).bfloat16()
g = run(N, Ho, Wo, C, K, dp, x, dyi)
print('dy=k: col c=0 tap0:', (g[:, 0, 0, 0] / (N * Ho * Wo)).tolist()[:40])
# pixel dependence: x = pixel linear index (interior), dy = delta at pixel j
x = torch.zeros(N, Ho + 2, Wo + 2, C)
for n in range(N):
    for p in range(Ho + 2):
        for q in range(Wo + 2):
            x[n, p, q, :] = (n * 100 + p * 10 + q)
x = x.bfloat16()
for pix in [(0, 0, 0), (0, 0, 5), (0, 3, 1), (1, 7, 7)]:
    dyi = torch.zeros(N, Ho, Wo, K); dyi[pix[0], pix[1], pix[2], :] = 1; dyi = dyi.bfloat16()
    g = run(N, Ho, Wo, C, K, dp, x, dyi)
    print('delta at', pix, 'taps ->', g[0, :, :, 0].flatten().tolist(), ' expect', [pix[0]*100 + (pix[1]+r)*10 + pix[2]+s for r in range(3) for s in range(3)])
