import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from keypoints_amd._lib import call, ptr
from keypoints_amd.runtime import precision_dtype, stream_ptr
z=np.load('/root/repo/tests/golden/layers.npz')
for precision in ('fp32','bf16','fp16'):
    tdt, code = precision_dtype(precision)
    x=torch.tensor(z['head.x'],device='cuda'); N,C,H,W=x.shape; K=5
    a=torch.empty(N,H,W,C,device='cuda',dtype=tdt); st=stream_ptr()
    call('kp_nchw_to_nhwc',ptr(x),ptr(a),N,C,C,H,W,0,code,st)
    wt=torch.tensor(z['head.weight'],device='cuda'); b=torch.tensor(z['head.bias'],device='cuda')
    out=torch.empty(N,K,H,W,device='cuda')
    call('kp_head_fwd',ptr(a),ptr(wt),ptr(b),0.01,ptr(out),N,H,W,C,K,code,st)
    gout=torch.tensor(z['head.w'],device='cuda')
    dxn=torch.full((N,H,W,C),float('nan'),device='cuda',dtype=tdt)
    dw=torch.zeros(K,C,device='cuda'); db=torch.zeros(K,device='cuda')
    call('kp_head_bwd',ptr(gout),ptr(out),ptr(a),ptr(wt),0.01,ptr(dxn),ptr(dw),ptr(db),None,N,H,W,C,K,code,st)
    got=dxn.float().permute(0,3,1,2).cpu().numpy(); ref=z['head.dx']
    err=np.abs(got-ref); i=np.unravel_index(err.argmax(),err.shape)
    print(precision,'dx maxerr',err.max(),'at',i,'got',got[i],'ref',ref[i],'refmax',np.abs(ref).max(), 'out there', out[i[0],:,i[2],i[3]].cpu().numpy(), 'refy', z['head.y'][i[0],:,i[2],i[3]])
    print('   y err', np.abs(out.cpu().numpy()-z['head.y']).max(), 'dw err', np.abs(dw.cpu().numpy().reshape(K,C,1,1)-z['head.dweight']).max(), np.abs(z['head.dweight']).max())
    # count of large errors
    print('   n bad', (err>0.02*np.abs(ref).max()).sum(), 'of', err.size)
