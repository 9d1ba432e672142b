#!/usr/bin/env python3
"""SURVEY 8 row f4 "at speed" (VERDICT r05 item 9): the keypoint bottleneck of TransporterNet.forward -- ssm + key2map of both frames
+ the combine step (/root/reference/keypoints/models/transporter.py:27-60) -- forward + backward, for every `combine_method`:
'max' as the ONE fused launch each way the train path uses (kp_keypoint_bottleneck_fwd / _bwd), 'max' / 'sum_and_clamp' / 'loop' as
the piecewise sequence (kp_spatial_softmax x 2, kp_gaussian x 2, kp_transport: five launches forward), and 'pretrained_network'
(point_map + the MAPPER Unit of mapper.py:50-52 + the blend).  At the bottleneck shapes of BASELINE config 4 (128 pairs, K = 10
heat-maps and 64 feature channels on 16 x 16) and config 2 (256 pairs, K = 4, 16 channels on 84 x 84).

    python tools/bench_f4.py [--reps 30]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from keypoints_amd.models import functional as MF
from keypoints_amd.models import knn, transporter

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=30)
args = ap.parse_args()
dev = torch.device('cuda', 0)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return 1e3 * t[len(t) // 2]   # median, us


print(f'{"shape":34s} {"combine_method":28s} {"fwd+bwd us":>11s} {"x fused max":>12s} {"alg. MB":>8s} {"GB/s":>7s}')
for name, B, K, C, H, W in (('config 4: 128 x (K=10, C=64) 16x16', 128, 10, 64, 16, 16), ('config 2: 256 x (K=4, C=16) 84x84', 256, 4, 16, 84, 84)):
    g = torch.Generator().manual_seed(0)
    heat_s = torch.randn(B, K, H, W, generator=g).to(dev)
    heat_t = torch.randn(B, K, H, W, generator=g).to(dev).requires_grad_(True)
    phi_s = torch.randn(B, C, H, W, generator=g).to(dev)
    phi_t = torch.randn(B, C, H, W, generator=g).to(dev).requires_grad_(True)
    gphi = torch.randn(B, C, H, W, generator=g).to(dev)
    ssm, key2map = knn.SpatialLogSoftmax(), knn.GaussianLike(sigma=0.1)
    mapper = transporter.make_mapper(K).to(dev).eval()
    for prm in mapper.parameters():
        prm.requires_grad_(False)
    masker = transporter.MaskMaker(mapper)
    # algorithmic bytes of the fused 'max' pair (fp32): forward reads 2 heat-maps + 2 feature maps, writes the blend + maps; the
    # backward reads the gradient + both feature maps, writes the two input gradients
    mb = 4.0 * B * H * W * ((2 * K + 2 * C) + (C + K) + (3 * C) + (C + K)) / 1e6

    def fused():
        heat_t.grad = phi_t.grad = None
        out = MF.keypoint_bottleneck(heat_s, heat_t, phi_s, phi_t, sigma=0.1)
        out[0].backward(gphi)

    def piecewise(mode, k2m):
        def run():
            heat_t.grad = phi_t.grad = None
            with torch.no_grad():
                k_s, _ = ssm(heat_s, probs=True)
                m_s = k2m(k_s, height=H, width=W)
            k_t, p_t = ssm(heat_t, probs=True)
            m_t = k2m(k_t, height=H, width=W)
            phi, _, _ = MF.transport(phi_s, phi_t, m_s, m_t, mode)
            phi.backward(gphi)
        return run

    base = timed(fused, args.reps)
    print(f'{name:34s} {"max (fused: 1 + 1 launches)":28s} {base:11.1f} {1.0:12.2f} {mb:8.1f} {mb / base * 1e3:7.0f}')
    for mode, k2m in (('max', key2map), ('sum_and_clamp', key2map), ('loop', key2map), ('pretrained_network', masker)):
        if mode == 'pretrained_network':
            # the frozen mapper gives no gradient to the keypoints (MaskMaker runs under no_grad, transporter.py:89-91)
            def run(mode=mode, k2m=k2m):
                phi_t.grad = None
                with torch.no_grad():
                    k_s, _ = ssm(heat_s, probs=True)
                    m_s = k2m(k_s, height=H, width=W)
                    k_t, _ = ssm(heat_t, probs=True)
                    m_t = k2m(k_t, height=H, width=W)
                phi, _, _ = MF.transport(phi_s, phi_t, m_s, m_t, mode)
                phi.backward(gphi)
            t = timed(run, args.reps)
        else:
            t = timed(piecewise(mode, k2m), args.reps)
        print(f'{"":34s} {mode + " (piecewise)":28s} {t:11.1f} {t / base:12.2f}')
