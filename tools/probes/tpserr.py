import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from keypoints_amd import data_augments as DA, tps as TPS
from oracle import kp_oracle as O
z=np.load('/root/repo/tests/golden/tps.npz')
for tag in 'ab':
    x=torch.tensor(z[f'{tag}.x']).cuda()
    th,ct,ro=[torch.tensor(z[f'{tag}.{k}0']) for k in ('theta','ctrl','rot')]
    e1=np.abs(TPS.tps_transform(x,th,ct).cpu().numpy()-z[f'{tag}.warp']).max()
    e2=np.abs(TPS.rotate_affine_grid_multi(x,ro).cpu().numpy()-z[f'{tag}.rot']).max()
    # fp64 reference noise floor of the reference itself
    w64=O.tps_transform(x.cpu().double(),th.double(),ct.double()).numpy()
    r64=O.rotate_affine_grid_multi(x.cpu().double(),ro.double()).numpy()
    print(tag,'hip-vs-ref32 tps',e1,'rot',e2,'| ref32-vs-ref64 tps',np.abs(z[f'{tag}.warp']-w64).max(),'rot',np.abs(z[f'{tag}.rot']-r64).max(), '| hip-vs-ref64', np.abs(TPS.tps_transform(x,th,ct).cpu().numpy()-w64).max())
