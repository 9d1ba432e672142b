#!/usr/bin/env python3
"""Latency of the inference-only keypoint extractor (keypoints_amd/inference.py): eager chain of C-ABI launches vs the
captured HIP graph, and the fused forms (Pong-class: the one-kernel Unit; model F: BatchNorm folded into each convolution's epilogue), per frame, for the Pong-class network (84x84 grey, K=4) and model F (128x128x3, K=10).
    python tools/bench_infer.py [--batch 1] [--frames 200]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from keypoints_amd.inference import KeypointExtractor
from keypoints_amd.models import transporter


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=1); ap.add_argument('--frames', type=int, default=200)
    args = ap.parse_args()
    for typ, cin, z, K, S, prec in (('VGG_PONG_LAYERNECK', 1, 16, 4, 84, 'fp32'), ('VGG_PONG_LAYERNECK', 1, 16, 4, 84, 'bf16'),
                                    ('F', 3, 64, 10, 128, 'bf16'), ('F', 3, 64, 10, 128, 'fp16')):
        torch.manual_seed(0)
        net = transporter.make(typ, cin, z, K).cuda().set_precision(prec)
        x = torch.rand(args.batch, cin, S, S, device='cuda')
        line = f'{typ:>18} {S}x{S} batch {args.batch} {prec}:'
        for graph, fused in ((False, False), (True, False), (False, 'auto'), (True, 'auto')):
            ex = KeypointExtractor(net, graph=graph, fused=fused)
            for _ in range(5):
                ex(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.frames):
                k = ex(x)
                k[0, 0, 0].item()          # the consumer reads the keypoints every frame
            dt = (time.perf_counter() - t0) / args.frames
            line += f'   {"fused " if fused else ""}{"graph" if graph else "eager"} {dt * 1e6:8.1f} us/frame'
        print(line)


if __name__ == '__main__':
    main()
