"""Can an HBM-bound BatchNorm pass run on the SAME CUs as the persistent weight-gradient kernel?  wgrad3x3_patch_k takes
206 VGPRs x 2 waves per SIMD and 113 KB of LDS: 96 VGPRs per SIMD and 44 KB stay free, enough for one 256-thread workgroup
of a lean (<= 96 VGPR) streaming kernel per CU.  Stand-in for it: bn_act_fwd_none_fast_k (73 VGPRs, no LDS).
MFMA lane = high-priority stream, BN lane = low-priority stream; both start together.
    python tools/probes/coresident.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr

N = 128; bf = torch.bfloat16; dev = 'cuda'
C, K, H = 256, 256, 64
a = (torch.randn(N, H + 2, H + 2, C, device=dev) * 0.5).to(bf)
dy = (torch.randn(N, H + 4, H + 4, K, device=dev) * 0.1).to(bf)
wd = (torch.randn(C, 3, 3, K, device=dev) * 0.05).to(bf)
gx = torch.empty(N, H + 2, H + 2, C, device=dev, dtype=bf)
parts = _lib.lib().kp_conv_wgrad_partials(N, H, H, C, K, 3, 2, 1)
scratch = torch.empty(parts, K, 3, 3, C, device=dev)
C2, H2 = 128, 128
y2 = torch.randn(N, H2, H2, C2, device=dev).to(bf)
out2 = torch.empty(N, H2 + 2, H2 + 2, C2, device=dev, dtype=bf)
mean = torch.zeros(C2, device=dev); inv = torch.ones(C2, device=dev); ga = torch.ones(C2, device=dev); be = torch.zeros(C2, device=dev)

hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)


def wgrad(st):
    call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 2, 1, st.cuda_stream)


def dgrad(st):
    call('kp_conv_dgrad', ptr(dy), ptr(wd), ptr(gx), N, H, H, K, C, 1, st.cuda_stream)


def bn(st):
    call('kp_bn_act_fwd', ptr(y2), ptr(mean), ptr(inv), ptr(ga), ptr(be), 0.01, 0, ptr(out2), 1, None, N, H2, H2, C2, 1, st.cuda_stream)


def timed(lanes, reps=5):
    """lanes: [(stream, [fn, ...])]; returns wall ms and per-lane ms (per repetition)."""
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
    timed([(hi, [wgrad, dgrad])]); timed([(lo, [bn])])
for name, mf in (('wgrad', wgrad), ('dgrad', dgrad)):
    tw, _ = timed([(hi, [mf])])
    tb, _ = timed([(lo, [bn])])
    print(f'{name} alone {tw * 1e3:.0f} us   BN pass alone {tb * 1e3:.0f} us')
    for nb in (1, 2):
        wall, lanes = timed([(hi, [mf]), (lo, [bn] * nb)])
        print(f'  {name} || {nb} BN pass(es): wall {wall * 1e3:.0f} us ({name} done at {lanes[0] * 1e3:.0f}, BN at {lanes[1] * 1e3:.0f})'
              f'   serial {1e3 * (tw + nb * tb):.0f} us   -> {100 * (1 - wall / (tw + nb * tb)):.1f} % saved')
    # the other launch order: BN enqueued first
    wall, lanes = timed([(lo, [bn]), (hi, [mf])])
    print(f'  BN enqueued first, then {name}: wall {wall * 1e3:.0f} us (BN at {lanes[0] * 1e3:.0f}, {name} at {lanes[1] * 1e3:.0f})')
