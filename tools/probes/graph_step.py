import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import keypoints_amd as ka
from keypoints_amd.models import transporter
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = sys.argv[2] if len(sys.argv) > 2 else 'F'
cin, z, K, size = (3, 64, 10, 128) if model == 'F' else (1, 16, 4, 84)
g = torch.Generator().manual_seed(3)
xs = torch.rand(B, cin, size, size, generator=g).cuda(); xt = torch.rand(B, cin, size, size, generator=g).cuda()


def build():
    torch.manual_seed(11)
    net = transporter.make(model, cin, z, K).cuda().set_precision('bf16')
    return net, ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)


def eager(net, opt, n):
    for _ in range(n):
        opt.zero_grad(); loss = ka.l2_reconstruction_loss(net(xs, xt)[0], xt); loss.backward(); opt.step()
    return loss


net, opt = build(); eager(net, opt, 5); torch.cuda.synchronize()
t0 = time.perf_counter(); l = eager(net, opt, 30); torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 30 * 1e3
net2, opt2 = build()
step = ka.graph.GraphedTrainStep(net2, opt2, None, (xs, xt), warmup=3)
for _ in range(2): step(xs, xt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): l2 = step(xs, xt)
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 30 * 1e3
print(f'{model} B={B}: eager {te:.3f} ms/step ({B / te * 1e3:.0f} pairs/s)   graph {tg:.3f} ms/step ({B / tg * 1e3:.0f} pairs/s)   losses {float(l):.5f} {float(l2):.5f}')
