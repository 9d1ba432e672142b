"""wgrad_l on the high-priority stream next to the BatchNorm backward of layer l-1 (reduce, finalize, apply) on a low-priority
stream -- the pairing the backward of a Unit offers (wgrad_l does not feed the dgrad_l -> BN_{l-1} -> dgrad_{l-1} chain).
    python tools/probes/coresident2.py [path/to/libkp_hip.so]   (a build whose BN backward kernels fit beside wgrad: <= 96 VGPRs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoints_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from keypoints_amd._lib import call, ptr

N = 128; bf = torch.bfloat16; dev = 'cuda'
if os.environ.get('KP_WALK'):
    call('kp_config', 5, int(os.environ['KP_WALK'], 0))
REDUCE_ONLY = os.environ.get('BN_LANE', 'full') == 'reduce'
hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)


def case(C, K, H, C2, H2):
    a = (torch.randn(N, H + 2, H + 2, C, device=dev) * 0.5).to(bf)
    dy = (torch.randn(N, H + 4, H + 4, K, device=dev) * 0.1).to(bf)
    parts = _lib.lib().kp_conv_wgrad_partials(N, H, H, C, K, 3, 2, 1)
    scratch = torch.empty(parts, K, 3, 3, C, device=dev)
    y2 = torch.randn(N, H2, H2, C2, device=dev).to(bf)
    g2 = (torch.randn(N, H2 + 2, H2 + 2, C2, device=dev) * 0.1).to(bf)
    dy2 = torch.empty(N, H2 + 4, H2 + 4, C2, device=dev, dtype=bf)
    mean = torch.zeros(C2, device=dev); inv = torch.ones(C2, device=dev); ga = torch.ones(C2, device=dev); be = torch.zeros(C2, device=dev)
    prows = _lib.lib().kp_bn_bwd_rows(1, 0, N, H2, H2, C2, 1)
    part = torch.empty(prows, 2, C2, device=dev); sums = torch.zeros(2 * C2, device=dev)
    ws = torch.empty(128 * C2, device=dev, dtype=torch.float64)
    dgam = torch.zeros(C2, device=dev); dbet = torch.zeros(C2, device=dev)

    def wgrad(st):
        call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 2, 1, st.cuda_stream)

    def bnb(st):
        s = st.cuda_stream
        call('kp_bn_act_bwd_reduce', ptr(g2), 1, 0, None, ptr(y2), ptr(mean), ptr(inv), ptr(ga), ptr(be), 0.01, ptr(part), N, H2, H2, C2, 1, s)
        call('kp_bn_bwd_finalize', ptr(part), ptr(ws), prows, C2, ptr(sums), ptr(dgam), ptr(dbet), s)
        if REDUCE_ONLY:
            return
        call('kp_bn_act_bwd_apply', ptr(g2), 1, 0, None, ptr(y2), ptr(mean), ptr(inv), ptr(ga), ptr(be), 0.01, ptr(sums), ptr(dy2), 2,
             N, H2, H2, C2, 1, s)

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

    timed([(hi, [wgrad])], 2); timed([(lo, [bnb])], 2)
    tw, _ = timed([(hi, [wgrad])])
    tb, _ = timed([(lo, [bnb])])
    ts, _ = timed([(hi, [wgrad, bnb])])
    wall, lanes = timed([(hi, [wgrad]), (lo, [bnb])])
    print(f'wgrad {C}>{K}@{H} {tw * 1e3:4.0f} us | BN bwd {C2}ch@{H2} {tb * 1e3:4.0f} us | one stream {ts * 1e3:4.0f} us | two streams {wall * 1e3:4.0f} us '
          f'(wgrad done {lanes[0] * 1e3:4.0f}, BN {lanes[1] * 1e3:4.0f})  -> {1e3 * (ts - wall):4.0f} us = {100 * (1 - wall / ts):.1f} % saved')


case(256, 256, 64, 128, 128)   # big BN layer beside a mid wgrad
case(256, 256, 64, 256, 64)    # the real pairing: wgrad_l and BN_{l-1} of the same resolution
case(128, 256, 64, 128, 64)
case(512, 512, 32, 512, 32)
case(64, 128, 128, 64, 128)
