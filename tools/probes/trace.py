import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from keypoints_amd._lib import call, ptr
N=128; st=torch.cuda.current_stream().cuda_stream; bf=torch.bfloat16
C,K,H = [int(v) for v in sys.argv[1:4]] if len(sys.argv)>3 else (256,128,128)
dbg = int(sys.argv[4]) if len(sys.argv)>4 else 0
call('kp_config',1,dbg)
a=(torch.randn(N,H+2,H+2,C,device='cuda')*0.5).to(bf); wf=(torch.randn(K,3,3,C,device='cuda')*0.05).to(bf)
bias=torch.zeros(K,device='cuda'); y=torch.empty(N,H,H,K,device='cuda',dtype=bf); stats=torch.empty(N*H*H//64,2,K,device='cuda')
tr=torch.zeros(512,device='cuda',dtype=torch.int64)
for i in range(2):
    call('kp_conv_fwd',ptr(a),ptr(wf),ptr(bias),ptr(y),ptr(stats),N,H+2,H+2,C,K,3,1,st)
call('kp_config_ptr',2,ptr(tr))
call('kp_conv_fwd',ptr(a),ptr(wf),ptr(bias),ptr(y),ptr(stats),N,H+2,H+2,C,K,3,1,st)
torch.cuda.synchronize()
call('kp_config_ptr',2,None)
t=tr.cpu().tolist()
t=[v for v in t if v]
d=[t[i+1]-t[i] for i in range(len(t)-1)]
per=4*9*(C//64)+3
print('stamps',len(t),'per tile',per)
for ti in range(0,min(len(d),3*per),per):
    print(f'dbg={dbg}', ' '.join(f'{v:5d}' for v in d[ti:ti+per][-23:]))
