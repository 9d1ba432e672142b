"""A/B of memory walk orders (kp_config(KP_CFG_WALK, mask)) inside ONE process on one box: the bench workload, `reps` rounds
of `steps` steps per mask, interleaved so that clock / thermal drift hits every arm alike.
    python tools/probes/walk_ab.py 0 1 3 7 15 ...   (masks; see kp_api.hip: g_kp_walk)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import keypoints_amd as ka
from keypoints_amd import _lib
from keypoints_amd.models import transporter
from keypoints_amd import unit_engine as ue

# each arm: WALK[:CHUNK_MELEMS]  (kp_config(KP_CFG_WALK) mask, unit_engine.BWD_CHUNK_ELEMS in units of 2^20 elements)
masks = sys.argv[1:] or ['0', '1']
steps, reps = int(os.environ.get('STEPS', 10)), int(os.environ.get('REPS', 3))
size, B, K = int(os.environ.get('SIZE', 128)), int(os.environ.get('BATCH', 128)), int(os.environ.get('KP', 10))
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = transporter.make('F', 3, 64, K).to(dev).set_precision(os.environ.get('PREC', 'bf16'))
opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
g = torch.Generator().manual_seed(1234)
xs = torch.rand(B, 3, size, size, generator=g).to(dev)
xt = torch.rand(B, 3, size, size, generator=g).to(dev)


def step():
    opt.zero_grad()
    out = net(xs, xt)
    loss = ka.l2_reconstruction_loss(out[0], xt)
    loss.backward()
    opt.step()
    return loss


for _ in range(5):
    step()
res = {m: [] for m in masks}
for r in range(reps):
    for m in masks:
        _lib.call('kp_config', 5, int(m.split(':')[0], 0))
        ue.BWD_CHUNK_ELEMS = (int(m.split(':')[1]) << 20) if ':' in m else 0
        step(); step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            loss = step()
        e1.record()
        torch.cuda.synchronize()
        res[m].append(e0.elapsed_time(e1) / steps)
for m in masks:
    print(f'walk {m:>8s}: ' + ' '.join(f'{t:7.3f}' for t in res[m]) + f'   min {min(res[m]):7.3f} ms/step   loss {float(loss):.6f}')
