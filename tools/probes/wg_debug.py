import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
from keypoints_amd.runtime import stream_ptr
for (N, Ho, Wo, C, K, dp) in [(2, 16, 16, 64, 64, 2), (2, 8, 8, 128, 128, 2), (1, 32, 32, 64, 128, 0), (2, 16, 16, 128, 64, 2), (4,16,16,256,512,2)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Ho + 2, Wo + 2, C, generator=g).bfloat16()
    dyi = torch.randn(N, Ho, Wo, K, generator=g).bfloat16()
    dy = torch.zeros(N, Ho + 2 * dp, Wo + 2 * dp, K, dtype=torch.bfloat16)
    dy[:, dp:dp + Ho, dp:dp + Wo] = dyi
    xf, df = x.double(), dyi.double()
    ref = torch.zeros(K, 3, 3, C, dtype=torch.float64)
    for r in range(3):
        for s in range(3):
            ref[:, r, s, :] = torch.einsum('npqk,npqc->kc', df, xf[:, r:r + Ho, s:s + Wo])
    scratch = torch.zeros(K, 3, 3, C, device='cuda')
    call('kp_conv_wgrad', ptr(x.cuda()), ptr(dy.cuda()), ptr(scratch), N, Ho, Wo, C, K, 3, dp, _lib.KP_BF16, stream_ptr())
    got = scratch.double().cpu()
    e = (got - ref).abs()
    print((N, Ho, Wo, C, K, dp), 'relL2', ((got - ref).norm() / ref.norm()).item(), 'bad frac', (e > 1e-2 * ref.abs().max()).double().mean().item())
    bad = (e > 1e-2 * ref.abs().max())
    if bad.any():
        print('  bad by k%64//8:', [bad[kk::8].double().mean().item() for kk in range(8)][:8])
        print('  bad by c:', [round(bad[:, :, :, cc].double().mean().item(),2) for cc in range(0, min(C,64), 4)])
        print('  bad by k:', [round(bad[kk].double().mean().item(),2) for kk in range(0, min(K,64), 4)])
        print('  bad by tap:', [round(bad[:, t // 3, t % 3].double().mean().item(),2) for t in range(9)])
