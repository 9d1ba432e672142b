"""Where the T-form bottleneck forward spends its time at the config-2 shape (256 pairs, 16 channels, K = 4, 84 x 84):
the kernel alone, whole / without phase A (heat-maps -> keypoints) / without phase B (render + blend), via kp_config(KP_CFG_DEBUG)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
from keypoints_amd.runtime import stream_ptr

B, C, K, H, W = 256, 16, 4, 84, 84
tdt, code = torch.bfloat16, _lib.KP_BF16
g = torch.Generator().manual_seed(0)
heat_s = (torch.randn(B, K, H, W, generator=g) * 3).cuda()
heat_t = (torch.randn(B, K, H, W, generator=g) * 3).cuda()
phi_s = torch.randn(B, H, W, C, generator=g).to(tdt).cuda()
phi_t = torch.randn(B, H, W, C, generator=g).to(tdt).cuda()
f32 = dict(device='cuda', dtype=torch.float32)
k = torch.empty(B, K, 2, **f32); p = torch.empty(B, K, H, **f32); q = torch.empty(B, K, W, **f32); m = torch.empty(B, K, H, W, **f32)
a = torch.empty(B, H + 2, W + 2, C, device='cuda', dtype=tdt)
ms = torch.empty(B, 1, H, W, **f32); mt = torch.empty(B, 1, H, W, **f32); arg = torch.empty(B, H, W, device='cuda', dtype=torch.uint8)
flush = torch.empty(512 << 20, device='cuda', dtype=torch.uint8)
st = stream_ptr()
L = _lib.lib()
for mask, name in ((0, 'whole'), (1, 'no phase A'), (2, 'no phase B'), (0, 'whole')):
    L.kp_config(1, mask)
    ts = []
    for it in range(12):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call('kp_keypoint_bottleneck_fwd_t', ptr(heat_s), ptr(heat_t), ptr(phi_s), ptr(phi_t), ptr(k), ptr(p), ptr(q), ptr(m), ptr(a), ptr(ms), ptr(mt),
             ptr(arg), B, C, K, H, W, 0.1, 1e-6, code, st)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    print(f'{name:12s} median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f}')
L.kp_config(1, 0)

# ---- backward
ga = (torch.randn(B, H + 2, W + 2, C, generator=g) * 0.1).to(tdt).cuda()
d_phi = torch.empty(B, H, W, C, device='cuda', dtype=tdt); d_heat = torch.empty(B, K, H, W, **f32)
call('kp_keypoint_bottleneck_fwd_t', ptr(heat_s), ptr(heat_t), ptr(phi_s), ptr(phi_t), ptr(k), ptr(p), ptr(q), ptr(m), ptr(a), ptr(ms), ptr(mt),
     ptr(arg), B, C, K, H, W, 0.1, 1e-6, code, st)
for mask, name in ((0, 'bwd whole'), (1, 'bwd no step 1'), (2, 'bwd no steps 2-3'), (0, 'bwd whole')):
    L.kp_config(1, mask)
    ts = []
    for it in range(12):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call('kp_keypoint_bottleneck_bwd_t', ptr(ga), None, None, ptr(phi_s), ptr(phi_t), ptr(ms), ptr(mt), ptr(arg), ptr(k), ptr(p), ptr(q),
             ptr(d_phi), ptr(d_heat), B, C, K, H, W, 0.1, 1e-6, code, st)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    print(f'{name:18s} median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f}')
L.kp_config(1, 0)
