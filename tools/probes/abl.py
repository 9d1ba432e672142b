import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_layers import timed
N=128; st=torch.cuda.current_stream().cuda_stream; bf=torch.bfloat16
for (C,K,H) in [(64,128,128),(256,128,128),(512,512,32)]:
    a=(torch.randn(N,H+2,H+2,C,device='cuda')*0.5).to(bf); wf=(torch.randn(K,3,3,C,device='cuda')*0.05).to(bf)
    bias=torch.zeros(K,device='cuda'); y=torch.empty(N,H,H,K,device='cuda',dtype=bf); stats=torch.empty(N*H*H//64,2,K,device='cuda')
    fl=2.0*N*H*H*9*C*K
    for mask in [int(m) for m in sys.argv[1:]] or [0,1,2,3]:
        # 256 / 512 are host-side: all-zero / constant inputs (data-dependent power); the rest is the kernel's mask (the measurement
        # instantiation runs whenever it is non-zero: 8 = that build without any ablation)
        if mask & 256: a.zero_(); wf.zero_()
        if mask & 512: a.fill_(0.37); wf.fill_(0.011)
        call('kp_config',1,mask & 255)
        t=timed(lambda: call('kp_conv_fwd',ptr(a),ptr(wf),ptr(bias),ptr(y),ptr(stats),N,H+2,H+2,C,K,3,1,st),5)
        print(f'C={C} K={K} H={H} dbg={mask}: {t*1e6:8.1f} us  {fl/t/1e12:7.0f} TF')
    call('kp_config',1,0)
