"""Per-step HIP-event times of the config-2 train step over 400 steps: which steps are slow, and is there a period?
python tools/probes/step_jitter.py [gc]   (gc: leave Python's collector on)"""
import gc, sys, time
import torch
sys.path.insert(0, '.')
import keypoints_amd as ka
from keypoints_amd.models import transporter

torch.manual_seed(0)
net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 4).cuda().set_precision('bf16')
opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
xs = torch.rand(256, 1, 84, 84, device='cuda') * 2 - 1
xt = torch.rand(256, 1, 84, 84, device='cuda') * 2 - 1


def step():
    opt.zero_grad()
    out = net(xs, xt)
    loss = ka.l2_reconstruction_loss(out[0], xt)
    loss.backward()
    opt.step()


for _ in range(20):
    step()
torch.cuda.synchronize()
if 'gc' not in sys.argv:
    gc.collect(); gc.freeze(); gc.disable()
N = 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
ev[0].record()
for i in range(N):
    t0 = time.perf_counter()
    step()
    host.append((time.perf_counter() - t0) * 1e3)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
srt = sorted(ms)
print('median', round(srt[N // 2], 4), 'mean', round(sum(ms) / N, 4), 'p90', round(srt[int(N * .9)], 4), 'p99', round(srt[int(N * .99)], 4), 'max', round(srt[-1], 4))
hs = sorted(host)
print('host ms per step: median', round(hs[N // 2], 3), 'p90', round(hs[int(N * .9)], 3), 'max', round(hs[-1], 3))
slow = [(i, round(ms[i], 3), round(host[i], 3)) for i in range(N) if ms[i] > 1.1 * srt[N // 2]]
print(len(slow), 'slow steps (> 1.1 x median): (index, gpu ms, host ms)', slow[:40])
