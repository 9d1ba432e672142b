import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import keypoints_amd as ka
from keypoints_amd.models import transporter
from oracle import kp_oracle as O
typ, cin, z, K, size, B = sys.argv[1], 3, 16, 4, int(sys.argv[2]), int(sys.argv[3])
plans = O.transporter_plans(typ, cin, z, K)
P, Bf = O.init_params(plans, seed=7)
g = torch.Generator().manual_seed(8)
xs = torch.rand(B, cin, size, size, generator=g); xt = torch.rand(B, cin, size, size, generator=g)
net = transporter.make(typ, cin, z, K); net.load_state_dict({**P, **Bf}, strict=True)
net = net.cuda().set_precision('fp32')
out = net(xs.cuda(), xt.cuda())
def rel(a, b): a, b = a.detach().double().cpu(), b.detach().double().cpu(); return float((a - b).norm() / b.norm())
for dt in (torch.float64, torch.float32):
    oP = {k: v.to(dt) for k, v in P.items()}; oB = {k: (v.to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in Bf.items()}
    with torch.no_grad():
        o = O.transporter_forward(plans, oP, xs.to(dt), xt.to(dt), oB)
    if dt == torch.float64: ref = o
    print(str(dt), 'x_t', rel(out[0], o[0]), 'phi', rel(out[1], o[1]), 'k', float((out[2].cpu().double() - o[2].double()).abs().max()), 'm', rel(out[3], o[3]), 'heat', rel(out[5], o[5]))
print('oracle32 vs 64: x_t', rel(o[0], ref[0]), 'phi', rel(o[1], ref[1]), 'heat', rel(o[5], ref[5]))
