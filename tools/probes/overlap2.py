"""Do the real kernels overlap when the chip is PARTITIONED with CU-masked streams?  (tools/probes/cumask.hip found:
mask bit i = CU i/8 of XCC i%8; 224 CUs sustain the same power-limited MFMA throughput as 256; 32 CUs stream 1.5 TB/s.)
MFMA lane: conv forward / weight gradient / data gradient of one layer on the upper CUs of every XCC (grid = its CU budget);
BN lane: BatchNorm forward / backward passes of another layer on the lower CUs.
    python tools/probes/overlap2.py [bn_cus ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr

hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(lo, hi):
    """Stream restricted to mask bits [lo, hi) = CUs lo/8 .. hi/8 - 1 of every XCC."""
    m = (ctypes.c_uint32 * 8)()
    for b in range(lo, hi):
        m[b >> 5] |= 1 << (b & 31)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


N = 128; bf = torch.bfloat16; dev = 'cuda'
# MFMA lane workload: 256 -> 256 @ 64x64 (forward with statistics, weight gradient, data gradient)
C, K, H = 256, 256, 64
a = (torch.randn(N, H + 2, H + 2, C, device=dev) * 0.5).to(bf)
wf = (torch.randn(K, 3, 3, C, device=dev) * 0.05).to(bf)
wd = (torch.randn(C, 3, 3, K, device=dev) * 0.05).to(bf)
bias = torch.zeros(K, device=dev)
y = torch.empty(N, H, H, K, device=dev, dtype=bf)
stats = torch.empty(N * H * H // 64, 2, K, device=dev)
dy = (torch.randn(N, H + 4, H + 4, K, device=dev) * 0.1).to(bf)
gx = torch.empty(N, H + 2, H + 2, C, device=dev, dtype=bf)
# BN lane workload: 128 channels @ 128x128 (forward apply; backward reduce + apply)
C2, H2 = 128, 128
y2 = torch.randn(N, H2, H2, C2, device=dev).to(bf)
out2 = torch.empty(N, H2 + 2, H2 + 2, C2, device=dev, dtype=bf)
g2 = (torch.randn(N, H2 + 2, H2 + 2, C2, device=dev) * 0.1).to(bf)
dy2 = torch.empty(N, H2 + 4, H2 + 4, C2, device=dev, dtype=bf)
mean = torch.zeros(C2, device=dev); inv = torch.ones(C2, device=dev); ga = torch.ones(C2, device=dev); be = torch.zeros(C2, device=dev)
part = torch.empty(N * H2, 2, C2, device=dev); sums = torch.zeros(2 * C2, device=dev)


def set_cus(n):
    call('kp_config', 6, n)


def mfma_work(st, cus):
    set_cus(cus)
    s = st.cuda_stream
    call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3, 1, s)
    parts = _lib.lib().kp_conv_wgrad_partials(N, H, H, C, K, 3, 2, 1)
    scratch = torch.empty(parts, K, 3, 3, C, device=dev)   # (allocated on the current stream; lives to the end of the probe)
    keep.append(scratch)
    call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 2, 1, s)
    call('kp_conv_dgrad', ptr(dy), ptr(wd), ptr(gx), N, H, H, K, C, 1, s)
    set_cus(0)


def bn_work(st, cus):
    set_cus(cus)
    s = st.cuda_stream
    call('kp_bn_act_fwd', ptr(y2), ptr(mean), ptr(inv), ptr(ga), ptr(be), 0.01, 0, ptr(out2), 1, None, N, H2, H2, C2, 1, s)
    call('kp_bn_act_bwd_reduce', ptr(g2), 1, 0, None, ptr(y2), ptr(mean), ptr(inv), ptr(ga), ptr(be), 0.01, ptr(part), N, H2, H2, C2, 1, s)
    call('kp_bn_act_bwd_apply', ptr(g2), 1, 0, None, ptr(y2), ptr(mean), ptr(inv), ptr(ga), ptr(be), 0.01, ptr(sums), ptr(dy2), 2,
         N, H2, H2, C2, 1, s)
    set_cus(0)


keep = []


def timed(fn_list, reps):
    """fn_list: [(fn, stream, cus, count)]; all lanes start together; returns (wall ms, [per-lane ms])."""
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    e0 = torch.cuda.Event(enable_timing=True); e0.record(main)
    ends = []
    for fn, st, cus, cnt in fn_list:
        st.wait_event(e0)
        for _ in range(reps * cnt):
            fn(st, cus)
        e = torch.cuda.Event(enable_timing=True); e.record(st); ends.append(e)
    for e in ends:
        main.wait_event(e)
    e1 = torch.cuda.Event(enable_timing=True); e1.record(main)
    torch.cuda.synchronize()
    keep.clear()
    return e0.elapsed_time(e1) / reps, [e0.elapsed_time(e) / reps for e in ends]


full = torch.cuda.Stream()
R = 6
for _ in range(2):
    timed([(mfma_work, full, 0, 1)], 2); timed([(bn_work, full, 0, 1)], 2)
t_m, _ = timed([(mfma_work, full, 0, 1)], R)
t_b, _ = timed([(bn_work, full, 0, 1)], R)
print(f'whole chip: MFMA set (fwd+wgrad+dgrad 256>256@64) {t_m:.3f} ms   BN set (fwd, bwd reduce, bwd apply 128ch@128) {t_b:.3f} ms')
for bn_cus in [int(v) for v in sys.argv[1:]] or [32, 48, 64]:
    s_bn, s_m = masked_stream(0, bn_cus), masked_stream(bn_cus, 256)
    mc = 256 - bn_cus
    tm_alone, _ = timed([(mfma_work, s_m, mc, 1)], R)
    tb_alone, _ = timed([(bn_work, s_bn, bn_cus, 1)], R)
    # how many BN sets fit beside one MFMA set?  try ratios around tm_alone / tb_alone
    best = None
    for cnt_b, cnt_m in ((1, 1), (1, 2), (1, 3), (2, 3), (1, 4)):
        wall, lanes = timed([(mfma_work, s_m, mc, cnt_m), (bn_work, s_bn, bn_cus, cnt_b)], 3)
        serial = cnt_m * t_m + cnt_b * t_b
        print(f'  bn {bn_cus:3d} CUs / mfma {mc}: {cnt_m} MFMA + {cnt_b} BN sets: wall {wall:.3f} ms (lanes {lanes[0]:.3f} / {lanes[1]:.3f})'
              f'   serial on the whole chip {serial:.3f} ms   -> {100 * (1 - wall / serial):5.1f} % saved')
    print(f'  alone on their partitions: MFMA set {tm_alone:.3f} ms ({mc} CUs)   BN set {tb_alone:.3f} ms ({bn_cus} CUs)')
