"""Weight gradient of the small-channel kernels on wide / odd shapes against an fp64 reference (which kernel takes them is the plan's business)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
bf=torch.bfloat16; st=torch.cuda.current_stream().cuda_stream
for (N,Ho,Wo,C,K) in [(1,8,300,16,32),(1,4,700,32,32),(2,5,513,32,16),(3,40,200,16,16),(1,3,1200,16,32),(2,130,130,32,32)]:
    torch.manual_seed(0)
    x=torch.randn(N,Ho+2,Wo+2,C,device='cuda').to(bf); dy=torch.randn(N,Ho,Wo,K,device='cuda').to(bf)
    parts=_lib.lib().kp_conv_wgrad_partials(N,Ho,Wo,C,K,3,0,1)
    xf, df = x.double().cpu(), dy.double().cpu()
    wref = torch.zeros(K, 3, 3, C, dtype=torch.float64)
    for r in range(3):
        for s_ in range(3):
            wref[:, r, s_, :] = torch.einsum('npqk,npqc->kc', df, xf[:, r:r + Ho, s_:s_ + Wo])
    s=torch.full((parts,K,3,3,C),float('nan'),device='cuda') if parts > 1 else torch.zeros(1,K,3,3,C,device='cuda')
    call('kp_conv_wgrad',ptr(x),ptr(dy),ptr(s),N,Ho,Wo,C,K,3,0,1,st)
    torch.cuda.synchronize()
    got=s.double().sum(0).cpu()
    print((N,Ho,Wo,C,K),'parts',parts,'nan',int(torch.isnan(got).sum()),'relerr',float((got-wref).norm()/wref.norm()))
