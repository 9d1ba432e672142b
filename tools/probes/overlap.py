"""Does an HBM-bound BN kernel overlap with the persistent MFMA conv kernel when launched on another stream?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
N=128; bf=torch.bfloat16
C,K,H=512,512,32
a=(torch.randn(N,H+2,H+2,C,device='cuda')*0.5).to(bf); wf=(torch.randn(K,3,3,C,device='cuda')*0.05).to(bf)
bias=torch.zeros(K,device='cuda'); y=torch.empty(N,H,H,K,device='cuda',dtype=bf); stats=torch.empty(N*H*H//64,2,K,device='cuda')
C2,H2=128,128
y2=(torch.randn(N,H2,H2,C2,device='cuda')).to(bf); out2=torch.empty(N,H2+2,H2+2,C2,device='cuda',dtype=bf)
mean=torch.zeros(C2,device='cuda'); inv=torch.ones(C2,device='cuda'); ga=torch.ones(C2,device='cuda'); be=torch.zeros(C2,device='cuda')
s1=torch.cuda.Stream(); s2=torch.cuda.Stream()
def conv(st): call('kp_conv_fwd',ptr(a),ptr(wf),ptr(bias),ptr(y),ptr(stats),N,H+2,H+2,C,K,3,1,st.cuda_stream)
def bn(st): call('kp_bn_act_fwd',ptr(y2),ptr(mean),ptr(inv),ptr(ga),ptr(be),0.01,0,ptr(out2),1,None,N,H2,H2,C2,1,st.cuda_stream)
def run(par, reps=10):
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(s1)
    s2.wait_stream(s1)
    for _ in range(reps):
        conv(s1)
        bn(s2 if par else s1); bn(s2 if par else s1)
    s1.wait_stream(s2)
    e1.record(s1)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
def only(f, reps=10):
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(s1)
    for _ in range(reps): f(s1)
    e1.record(s1); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
for _ in range(2): run(0); run(1)
if len(sys.argv) > 1: call('kp_config', 1, int(sys.argv[1]))
print('conv alone %.3f ms, bn alone %.3f ms' % (only(conv), only(bn)))
print('sequential (conv + 2 bn): %.3f ms' % run(0))
print('two streams             : %.3f ms' % run(1))
