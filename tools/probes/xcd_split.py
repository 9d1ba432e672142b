#!/usr/bin/env python3
"""Go / no-go (round-4 verdict item 3c): the two encoder chains of a model-F step (feature / keypoint: forward + backward, the
part of the step where two independent chains exist) on two HALVES of the chip, so that one half's convolutions meet the other
half's BatchNorm passes instead of both chains fighting for the same CUs.
  plain       : two ordinary streams, persistent grids of 256 (what the step does today, `chain` side stream)
  xcd 4+4     : chain A on XCDs 0-3, chain B on XCDs 4-7 (hipExtStreamCreateWithCUMask), persistent grids of 128
  half of each: chain A on CUs 0-15 of every XCD, chain B on CUs 16-31, persistent grids of 128
  serial      : one stream, grids of 256
Mask bit i = CU i / 8 of XCC i % 8 (tools/probes/cumask.hip).  The weight-gradient side streams are switched off in every arm
that uses masked streams (a side stream of a masked stream would be an unmasked one) and, for a like-for-like row, in a second
plain arm.    python tools/probes/xcd_split.py [precision=fp16] [batch=128]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import keypoints_amd as ka
from keypoints_amd import _lib, _probe, runtime
from keypoints_amd.models import transporter
from keypoints_amd.unit_engine import shared_packs

hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(bits):
    m = (ctypes.c_uint32 * 8)()
    for b in bits:
        m[b >> 5] |= 1 << (b & 31)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = transporter.make('F', 3, 64, 10).to(dev).set_precision(prec)
x = torch.rand(B, 3, 128, 128, device=dev)


def chains(sa, sb):
    cur = torch.cuda.current_stream()
    with shared_packs():
        if sa is None:
            a = net.feature(x); b = net.keypoint(x)
        else:
            sa.wait_stream(cur); sb.wait_stream(cur)
            with torch.cuda.stream(sa):
                a = net.feature(x)
            with torch.cuda.stream(sb):
                b = net.keypoint(x)
            cur.wait_stream(sa); cur.wait_stream(sb)
    (a.sum() + b.sum()).backward()   # each Unit's backward runs on the stream of its forward
    if sa is not None:
        cur.wait_stream(sa); cur.wait_stream(sb)


def timed(sa, sb, n=8):
    for _ in range(3):
        chains(sa, sb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        chains(sa, sb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def cus(n):
    _lib.call('kp_config', _lib.KP_CFG_CUS, n)


plain = (torch.cuda.Stream(), torch.cuda.Stream())
xcd = (masked_stream([i for i in range(256) if i % 8 < 4]), masked_stream([i for i in range(256) if i % 8 >= 4]))
half = (masked_stream(range(0, 128)), masked_stream(range(128, 256)))
for rnd in range(3):
    runtime._disabled_roles.clear()
    cus(0)
    t_serial = timed(None, None)
    t_plain_w = timed(*plain)
    _probe.disable_roles('wgrad')
    t_serial_nw = timed(None, None)
    t_plain = timed(*plain)
    cus(128)
    t_plain128 = timed(*plain)
    t_xcd = timed(*xcd)
    t_half = timed(*half)
    cus(0)
    print(f'round {rnd}: serial {t_serial:.3f} (no wgrad stream {t_serial_nw:.3f}) | two plain streams {t_plain_w:.3f} (no wgrad stream {t_plain:.3f}; '
          f'grids of 128: {t_plain128:.3f}) | XCDs 0-3 / 4-7 {t_xcd:.3f} | lower / upper half of every XCD {t_half:.3f}   ms per fwd+bwd of both encoders', flush=True)
