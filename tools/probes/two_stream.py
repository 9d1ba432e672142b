#!/usr/bin/env python3
"""Probe: do the independent encoder chains of a Transporter step gain from running on two HIP streams?
(the tails / statistics-finalize gaps of one chain filled by the other's kernels).  Forward-only and forward+backward of
net.feature / net.keypoint on the same input, one stream vs two.  python tools/probes/two_stream.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import keypoints_amd as ka
from keypoints_amd.models import transporter
from keypoints_amd.unit_engine import shared_packs

dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = transporter.make('F', 3, 64, 10).to(dev).set_precision('bf16')
B = int(os.environ.get('B', 128))
x = torch.rand(B, 3, 128, 128, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
if os.environ.get('PRIO'):
    s1, s2 = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)


def fwd_serial():
    with torch.no_grad(), shared_packs():
        a = net.feature(x); b = net.keypoint(x)
    return a, b


def fwd_two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.no_grad():
        with torch.cuda.stream(s1):
            a = net.feature(x)
        with torch.cuda.stream(s2):
            b = net.keypoint(x)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b


def fb_serial():
    a = net.feature(x); b = net.keypoint(x)
    (a.sum() + b.sum()).backward()


def fb_two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = net.feature(x)
    with torch.cuda.stream(s2):
        b = net.keypoint(x)
    cur.wait_stream(s1); cur.wait_stream(s2)
    (a.sum() + b.sum()).backward()   # autograd runs each Unit's backward on the stream of its forward


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rnd in range(2):
    print(f'forward  : one stream {timed(fwd_serial):7.3f} ms   two streams {timed(fwd_two):7.3f} ms')
    print(f'fwd + bwd: one stream {timed(fb_serial):7.3f} ms   two streams {timed(fb_two):7.3f} ms')
