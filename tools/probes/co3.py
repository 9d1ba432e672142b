"""How fast does a streaming BatchNorm pass run BESIDE a persistent matrix-core kernel, as a function of the bytes it keeps
in flight per thread?  Data gradient (179 VGPRs x 2 waves / SIMD: 144 VGPRs and 11 KB of LDS free per SIMD / CU) on one stream,
bn_act_fwd_none (4-channel variant, KP_UNR vectors in flight per thread) on another.   python tools/probes/co3.py [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoints_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from keypoints_amd._lib import call, ptr

N = 128; bf = torch.bfloat16; dev = 'cuda'
C, K, H = 256, 256, 64
dy = (torch.randn(N, H + 4, H + 4, K, device=dev) * 0.1).to(bf)
wd = (torch.randn(C, 3, 3, K, device=dev) * 0.05).to(bf)
gx = torch.empty(N, H + 2, H + 2, C, device=dev, dtype=bf)
C2, H2 = 256, 64
y2 = torch.randn(N, H2, H2, C2, device=dev).to(bf)
out2 = torch.empty(N, H2 + 2, H2 + 2, C2, device=dev, dtype=bf)
mean = torch.zeros(C2, device=dev); inv = torch.ones(C2, device=dev); ga = torch.ones(C2, device=dev); be = torch.zeros(C2, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def dgrad(st):
    call('kp_conv_dgrad', ptr(dy), ptr(wd), ptr(gx), N, H, H, K, C, 1, st.cuda_stream)


def bn(st):
    call('kp_bn_act_fwd', ptr(y2), ptr(mean), ptr(inv), ptr(ga), ptr(be), 0.01, 0, ptr(out2), 1, None, N, H2, H2, C2, 1, st.cuda_stream)


def timed(lanes, reps=6):
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    tot, per = 0.0, [0.0] * len(lanes)
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e0.record(main)
        ends = []
        for st, fns in lanes:
            st.wait_event(e0)
        for st, fns in lanes:
            for f in fns:
                f(st)
            e = torch.cuda.Event(enable_timing=True); e.record(st); ends.append(e)
        for e in ends:
            main.wait_event(e)
        e1 = torch.cuda.Event(enable_timing=True); e1.record(main)
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
        for i, e in enumerate(ends):
            per[i] += e0.elapsed_time(e)
    return tot / reps, [p / reps for p in per]


for _ in range(2):
    timed([(sa, [dgrad])]); timed([(sb, [bn])])
tw, _ = timed([(sa, [dgrad] * 2)]); tw /= 2
tb, _ = timed([(sb, [bn] * 4)]); tb /= 4
gb = (y2.numel() + out2.numel()) * 2 / 1e9
print(f'dgrad alone {tw * 1e3:.0f} us   BN pass alone {tb * 1e3:.0f} us ({gb / tb * 1e3:.0f} GB/s)')
for nd, nb in ((2, 1), (2, 2), (2, 4), (2, 6)):
    wall, lanes = timed([(sa, [dgrad] * nd), (sb, [bn] * nb)])
    print(f'  {nd} dgrad || {nb} BN: wall {wall * 1e3:.0f} us (dgrad lane done at {lanes[0] * 1e3:.0f}, BN lane at {lanes[1] * 1e3:.0f})'
          f'   serial {1e3 * (nd * tw + nb * tb):.0f} us   -> {100 * (1 - wall / (nd * tw + nb * tb)):.1f} % saved')
