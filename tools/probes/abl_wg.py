import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from keypoints_amd._lib import call, ptr
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_layers import timed
N=128; st=torch.cuda.current_stream().cuda_stream; bf=torch.bfloat16
for (C,K,H) in [(256,256,64),(512,512,32),(512,512,16),(64,128,128)]:
    a=(torch.randn(N,H+2,H+2,C,device='cuda')*0.5).to(bf)
    dy=torch.zeros(N,H+4,H+4,K,device='cuda',dtype=bf); dy[:,2:-2,2:-2]=(torch.randn(N,H,H,K,device='cuda')*0.5).to(bf)
    from keypoints_amd import _lib as L
    scratch=torch.zeros(L.lib().kp_conv_wgrad_partials(N,H,H,C,K,3,2,1),K,3,3,C,device='cuda')
    fl=2.0*N*H*H*9*C*K
    for mask in [0,4]:
        call('kp_config',1,mask)
        t=timed(lambda: call('kp_conv_wgrad',ptr(a),ptr(dy),ptr(scratch),N,H,H,C,K,3,2,1,st),5)
        print(f'C={C} K={K} H={H} dbg={mask}: {t*1e6:8.1f} us  {fl/t/1e12:7.0f} TF')
    call('kp_config',1,0)
