"""Host-side enqueue time of a config-2 train step (no sync inside the step) against the GPU's step time: is the launch-issuing thread ahead?
python tools/probes/host_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import keypoints_amd as ka
from keypoints_amd.models import transporter
torch.manual_seed(0)
net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 4).cuda().set_precision('bf16')
opt = ka.FusedAdam(net.parameters(), lr=1e-3).bind(net)
xs = (torch.rand(256, 1, 84, 84) * 2 - 1).cuda(); xt = (torch.rand(256, 1, 84, 84) * 2 - 1).cuda()
def step(t):
    t0 = time.perf_counter(); opt.zero_grad()
    t1 = time.perf_counter(); out = net(xs, xt); loss = ka.l2_reconstruction_loss(out[0], xt)
    t2 = time.perf_counter(); loss.backward()
    t3 = time.perf_counter(); opt.step()
    t4 = time.perf_counter()
    if t is not None:
        t.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
for _ in range(10): step(None)
ka.runtime.freeze_gc()
torch.cuda.synchronize()
T = []
w0 = time.perf_counter()
for _ in range(50): step(T)
h1 = time.perf_counter()
torch.cuda.synchronize()
w1 = time.perf_counter()
import numpy as np
T = np.array(T) * 1e3
print('host enqueue per step (ms): zero_grad %.3f forward+loss %.3f backward %.3f optimizer %.3f  total %.3f' % (*T.mean(0), T.sum(1).mean()))
print('wall per step %.3f ms; host finished enqueueing %.3f ms before the GPU finished (over 50 steps)' % ((w1 - w0) / 50 * 1e3, (w1 - h1) * 1e3))
