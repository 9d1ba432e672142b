#!/usr/bin/env python3
"""Probe: two identical Unit chains (feature / keypoint encoder, forward + backward) on two streams fall into lockstep
(both convolutions at once, then both BatchNorm passes at once: nothing overlaps).  Does an initial offset between the
chains (a sleep kernel in front of one) keep them out of phase, so that one chain's BatchNorm passes run beside the other's
matrix-core kernels?  python tools/probes/stagger.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import keypoints_amd as ka
from keypoints_amd.models import transporter

dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = transporter.make('F', 3, 64, 10).to(dev).set_precision('bf16')
x = torch.rand(128, 3, 128, 128, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
CYC_PER_US = 100   # torch.cuda._sleep counts in ~10 ns ticks on this device (calibrated below)


def run(delay_fwd_us, delay_bwd_us):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = net.feature(x)
    with torch.cuda.stream(s2):
        if delay_fwd_us:
            torch.cuda._sleep(int(delay_fwd_us * CYC_PER_US))
        b = net.keypoint(x)
    with torch.cuda.stream(s1):
        la = a.sum()
    with torch.cuda.stream(s2):
        lb = b.sum()
        if delay_bwd_us:
            torch.cuda._sleep(int(delay_bwd_us * CYC_PER_US))
    with torch.cuda.stream(s1):
        la.backward()
    with torch.cuda.stream(s2):
        lb.backward()
    cur.wait_stream(s1); cur.wait_stream(s2)


def timed(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# calibrate the sleep
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(1000000); e1.record(); torch.cuda.synchronize()
CYC_PER_US = 1000000 / (e0.elapsed_time(e1) * 1e3)
print('sleep ticks per us', CYC_PER_US)
for df, db in [(0, 0), (150, 0), (0, 150), (150, 150), (0, 300), (300, 300), (0, 600), (0, 0)]:
    print(f'delay fwd {df:4d} us, bwd {db:4d} us: {timed(lambda: run(df, db)):7.3f} ms')
