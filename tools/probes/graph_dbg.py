import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import keypoints_amd as ka
from keypoints_amd import runtime
from keypoints_amd.models import transporter
runtime._disabled_roles.update(sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] else [])
g = torch.Generator().manual_seed(3)
xs = torch.rand(8, 3, 128, 128, generator=g).cuda(); xt = torch.rand(8, 3, 128, 128, generator=g).cuda()
torch.manual_seed(11)
net = transporter.make('F', 3, 64, 10).cuda().set_precision('bf16')
opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
step = ka.graph.GraphedTrainStep(net, opt, None, (xs, xt), warmup=2)
print('captured with roles disabled:', runtime._disabled_roles, float(step(xs, xt)))
