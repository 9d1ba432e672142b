"""Stream priorities for the model-F step: the main chain on a HIGH-priority stream (its side streams keep the default priority) against
everything at the default priority.  python tools/probes/prio_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import keypoints_amd as ka
from keypoints_amd.models import transporter

dev = torch.device('cuda', 0)
B, size, K = 128, 128, 10
torch.manual_seed(0)
net = transporter.make('F', 3, 64, K).to(dev).set_precision('fp16')
opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
opt.scaler = ka.amp.LossScaler('cuda')
g = torch.Generator().manual_seed(1234)
xs = torch.rand(B, 3, size, size, generator=g).to(dev)
xt = torch.rand(B, 3, size, size, generator=g).to(dev)


def step():
    opt.zero_grad()
    out = net(xs, xt)
    loss = ka.l2_reconstruction_loss(out[0], xt)
    with ka.amp.scale_loss(loss, opt) as scaled:
        scaled.backward()
    opt.step()
    return loss


# arm 'sides-high': the side streams (weight gradients, keypoint chain, ring) at HIGH priority, the main chain at the default one
from keypoints_amd import runtime as _rt
_orig_side = _rt.side_stream
_SIDE_PRIO = [0]


def _side(role):
    cur = torch.cuda.current_stream()
    key = (cur.device.index, cur.stream_id, role, _SIDE_PRIO[0])
    s_ = _rt._side_streams.get(key)
    if s_ is None:
        s_ = _rt._side_streams[key] = torch.cuda.Stream(device=cur.device, priority=_SIDE_PRIO[0])
        _rt._parents[s_] = cur
    _rt._used.add(s_)
    return s_


_rt.side_stream = _side
import keypoints_amd.unit_engine as _ue
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
print('priority range', lo, hi)
streams = {'default': None, 'main-high': torch.cuda.Stream(priority=-1), 'sides-high': None}
for name, st in streams.items():
    _SIDE_PRIO[0] = -1 if name == 'sides-high' else 0
    ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(6):
            step()
torch.cuda.synchronize()
for rep in range(3):
    for name, st in streams.items():
        _SIDE_PRIO[0] = -1 if name == 'sides-high' else 0
        ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            step(); step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                step()
            e1.record()
            torch.cuda.synchronize()
            print(f'{name:16s} {e0.elapsed_time(e1) / 20:7.3f} ms/step')
