#!/usr/bin/env python3
"""Probe: two Unit chains (feature / keypoint encoder, forward + backward) on two streams with their matrix-core launches
SERIALISED in strict alternation (A.conv, B.conv, A.conv, ...: each waits for the previous one's event), so that one chain's
BatchNorm passes always run beside the other chain's matrix-core kernel instead of both chains moving in lockstep.
The interleaved enqueue order is emulated with two Python threads and a turn lock around the MFMA-class calls.
python tools/probes/mfma_token.py [lib.so]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from keypoints_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import keypoints_amd as ka
from keypoints_amd import unit_engine
from keypoints_amd.models import transporter

sys.setswitchinterval(2e-5)   # the turn lock hands over between two Python threads: do not wait 5 ms for the GIL
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = transporter.make('F', 3, 64, 10).to(dev).set_precision('bf16')
x = torch.rand(128, 3, 128, 128, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
MFMA = {'kp_conv_fwd', 'kp_conv_dgrad', 'kp_conv_wgrad'}

cv = threading.Condition()
state = dict(turn=0, alive=[True, True], token=None, on=False)
tls = threading.local()
orig_call = unit_engine.call


def call(name, *a):
    if not state['on'] or name not in MFMA or not hasattr(tls, 'me'):
        return orig_call(name, *a)
    me = tls.me
    with cv:
        while state['turn'] != me and state['alive'][1 - me]:
            cv.wait()
        st = torch.cuda.current_stream()
        if state['token'] is not None:
            st.wait_event(state['token'])
        r = orig_call(name, *a)
        ev = torch.cuda.Event(); ev.record(st); state['token'] = ev
        state['turn'] = 1 - me
        cv.notify_all()
    return r


unit_engine.call = call


def chain(me, unit, stream, fwd_only):
    tls.me = me
    torch.cuda.set_device(0)
    with torch.cuda.stream(stream):
        if fwd_only:
            with torch.no_grad():
                unit(x)
        else:
            unit(x).sum().backward()
    with cv:
        state['alive'][me] = False
        cv.notify_all()


def run(token, fwd_only=False):
    state.update(turn=0, alive=[True, True], token=None, on=token)
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur); sB.wait_stream(cur)
    ta = threading.Thread(target=chain, args=(0, net.feature, sA, fwd_only))
    tb = threading.Thread(target=chain, args=(1, net.keypoint, sB, fwd_only))
    ta.start(); tb.start(); ta.join(); tb.join()
    cur.wait_stream(sA); cur.wait_stream(sB)
    from keypoints_amd import runtime
    runtime.join_side_streams()


def timed(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    global cpu_ms
    cpu_ms = (t1 - t0) / n * 1e3     # enqueue time: must stay below the GPU time for the result to mean anything
    return (time.perf_counter() - t0) / n * 1e3


for rnd in range(2):
    print(f'forward only: lockstep {timed(lambda: run(False, True)):7.3f} ms   alternating MFMA token {timed(lambda: run(True, True)):7.3f} ms')
    a = timed(lambda: run(False)); ca = cpu_ms; b = timed(lambda: run(True)); cb = cpu_ms
    print(f'fwd + bwd   : lockstep {a:7.3f} ms (enqueue {ca:.2f})   alternating MFMA token {b:7.3f} ms (enqueue {cb:.2f})')
