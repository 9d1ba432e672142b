#!/usr/bin/env python3
"""Calibration run for tests/test_gpu_fullsize.py: one full train step of each BASELINE configuration at its
full per-GPU batch, HIP path (fp32 strict and bf16) against the oracle run on the GPU in fp32 (the oracle is
plain torch and device-agnostic).  Prints one JSON line per case with every error measure the tests bound."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import keypoints_amd as ka
from keypoints_amd.models import transporter
from oracle import kp_oracle as O


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def run(tag, typ, cin, z, K, size, B, precision, pm1=False, steps=1):
    dev = 'cuda'
    plans = O.transporter_plans(typ, cin, z, K)
    P, Bf = O.init_params(plans, seed=3)
    net = transporter.make(typ, cin, z, K)
    net.load_state_dict({**P, **Bf}, strict=True)
    net = net.to(dev).set_precision(precision)
    opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
    g = torch.Generator().manual_seed(21)
    xs = torch.rand(B, cin, size, size, generator=g)
    xt = torch.rand(B, cin, size, size, generator=g)
    if pm1:
        xs, xt = xs * 2 - 1, xt * 2 - 1
    xs, xt = xs.to(dev), xt.to(dev)
    oP = {k: v.to(dev) for k, v in P.items()}
    oB = {k: v.to(dev) for k, v in Bf.items()}
    state = {}
    rows = []
    for s in range(steps):
        t0 = time.time()
        opt.zero_grad()
        out = net(xs, xt)
        loss = ka.l2_reconstruction_loss(out[0], xt)
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        opt.step()
        torch.cuda.synchronize()
        t1 = time.time()
        oloss, oout, oG = O.transporter_train_step(plans, oP, oB, state, xs, xt)
        torch.cuda.synchronize()
        t2 = time.time()
        r = dict(tag=tag, precision=precision, step=s, B=B, hip_s=t1 - t0, oracle_s=t2 - t1,
                 loss=float(loss.item()), oloss=float(oloss.item()),
                 loss_rel=abs(float(loss.item()) - float(oloss.item())) / abs(float(oloss.item())),
                 xt_max=float(oout[0].abs().max()), xt_err=float((out[0] - oout[0]).abs().max()),
                 xt_rel=rel(out[0], oout[0]),
                 k_err=float((out[2] - oout[2]).abs().max()), m_err=float((out[3] - oout[3]).abs().max()),
                 heat_rel=rel(out[5], oout[5]), phi_rel=rel(out[1], oout[1]))
        if s == 0:
            gr = {}
            for n, gv in grads.items():
                og = oG[n]
                gr[n] = (rel(gv, og), float(og.norm()), float(gv.abs().max()))
            names = sorted(gr, key=lambda n: -gr[n][0])
            r['grad_worst'] = [(n, gr[n]) for n in names[:12]]
            r['grad_median'] = float(np.median([v[0] for n, v in gr.items() if v[1] > 0]))
            sd = net.state_dict()
            perr = {n: rel(sd[n], oP[n]) for n in oP}
            berr = {n: float((sd[n].double() - oB[n].double()).abs().max() / max(1.0, float(oB[n].double().abs().max())))
                    for n in oB}
            r['param_worst'] = sorted(perr.items(), key=lambda kv: -kv[1])[:4]
            r['buf_worst'] = sorted(berr.items(), key=lambda kv: -kv[1])[:4]
            # Adam step direction: fraction of parameters whose update sign differs
        rows.append(r)
        print(json.dumps(r), flush=True)
    return rows


if __name__ == '__main__':
    which = sys.argv[1:] or ['c4f32', 'c4bf16', 'c2bf16', 'c2f32', 'c5bf16']
    torch.backends.cudnn.benchmark = False
    for w in which:
        if w == 'c4f32':
            run('c4', 'F', 3, 64, 10, 128, 128, 'fp32')
        elif w == 'c4bf16':
            run('c4', 'F', 3, 64, 10, 128, 128, 'bf16')
        elif w == 'c4traj':
            run('c4traj', 'F', 3, 64, 10, 128, 128, 'bf16', steps=20)
        elif w == 'c2bf16':
            run('c2', 'VGG_PONG_LAYERNECK', 1, 16, 4, 84, 256, 'bf16', pm1=True)
        elif w == 'c2f32':
            run('c2', 'VGG_PONG_LAYERNECK', 1, 16, 4, 84, 256, 'fp32', pm1=True)
        elif w == 'c5bf16':
            run('c5', 'F', 3, 64, 30, 256, int(os.environ.get('C5B', 64)), 'bf16')
        elif w == 'c5f32':
            run('c5', 'F', 3, 64, 30, 256, int(os.environ.get('C5B', 16)), 'fp32')
