#!/usr/bin/env python3
"""Calibration 2: fp64 oracle on the GPU (MIOpen off -> torch's native im2col+GEMM convolution, exact fp32/fp64),
noise floor of the fp32 oracle itself, bf16 error at random init and at warmed-up weights, 20-step trajectories."""
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

torch.backends.cudnn.enabled = False


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def oracle_step(plans, P, Bf, xs, xt, dtype, state=None):
    oP = {k: v.to('cuda', dtype).clone() for k, v in P.items()}
    oB = {k: (v.to('cuda', dtype) if v.dtype.is_floating_point else v.to('cuda')).clone() for k, v in Bf.items()}
    torch.cuda.synchronize(); t0 = time.time()
    loss, out, G = O.transporter_train_step(plans, oP, oB, {} if state is None else state, xs.to(dtype), xt.to(dtype))
    torch.cuda.synchronize()
    return loss, out, G, oP, oB, time.time() - t0


def hip_step(net, opt, xs, xt):
    opt.zero_grad()
    out = net(xs, xt)
    loss = ka.l2_reconstruction_loss(out[0], xt)
    with ka.amp.scale_loss(loss, opt) as scaled:
        scaled.backward()
    sc = opt.scaler.scale if opt.scaler is not None else 1.0
    G = {n: p.grad.detach().clone() / sc for n, p in net.named_parameters()}
    opt.step()
    return loss.detach(), out, G


def make(typ, cin, z, K, P, Bf, precision):
    net = transporter.make(typ, cin, z, K)
    net.load_state_dict({**{k: v.float().cpu() for k, v in P.items()}, **{k: v.cpu() for k, v in Bf.items()}}, strict=True)
    net = net.to('cuda').set_precision(precision)
    opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
    if precision == 'fp16':
        opt.scaler = ka.amp.LossScaler('cuda', init_scale=float(os.environ.get('SCALE', 65536)))
    return net, opt


def compare(tag, out, loss, G, rout, rloss, rG, noise):
    gerr = {n: rel(G[n], rG[n]) for n in G if n not in noise}
    worst = sorted(gerr.items(), key=lambda kv: -kv[1])[:5]
    r = dict(tag=tag, loss=float(loss), rloss=float(rloss), loss_rel=abs(float(loss) - float(rloss)) / abs(float(rloss)),
             xt_max=float(rout[0].abs().max()), xt_err=float((out[0].double() - rout[0].double()).abs().max()),
             xt_rel=rel(out[0], rout[0]), k_err=float((out[2].double() - rout[2].double()).abs().max()),
             m_err=float((out[3].double() - rout[3].double()).abs().max()), heat_rel=rel(out[5], rout[5]),
             phi_rel=rel(out[1], rout[1]), g_median=float(np.median(list(gerr.values()))),
             g_max=float(max(gerr.values())), g_worst=[(n, round(e, 5)) for n, e in worst])
    print(json.dumps(r), flush=True)
    return r


def noise_names(plans):
    return {f"{n}.{s['key']}.bias" for n, pl in plans.items() for s in pl if s['op'] == 'conv3'}


def main(which):
    typ, cin, z, K, size, B = dict(c4=('F', 3, 64, 10, 128, 128), c5=('F', 3, 64, 30, 256, int(os.environ.get('C5B', 32))),
                                   c2=('VGG_PONG_LAYERNECK', 1, 16, 4, 84, 256))[which]
    plans = O.transporter_plans(typ, cin, z, K)
    P, Bf = O.init_params(plans, seed=3)
    noise = noise_names(plans)
    g = torch.Generator().manual_seed(21)
    xs = torch.rand(B, cin, size, size, generator=g)
    xt = torch.rand(B, cin, size, size, generator=g)
    if which == 'c2':
        xs, xt = xs * 2 - 1, xt * 2 - 1
    xs, xt = xs.cuda(), xt.cuda()

    r64 = oracle_step(plans, P, Bf, xs, xt, torch.float64)
    print(json.dumps(dict(tag='oracle64_time', s=r64[5])), flush=True)
    r32 = oracle_step(plans, P, Bf, xs, xt, torch.float32)
    print(json.dumps(dict(tag='oracle32_time', s=r32[5])), flush=True)
    compare(which + ' oracle32 vs oracle64', r32[1], r32[0], r32[2], r64[1], r64[0], r64[2], noise)
    for prec in os.environ.get('PRECS', 'fp32,bf16,fp16').split(','):
        net, opt = make(typ, cin, z, K, P, Bf, prec)
        loss, out, G = hip_step(net, opt, xs, xt)
        if opt.scaler is not None:
            print(json.dumps(dict(tag='scaler', scale=opt.scaler.scale, skipped=opt.scaler.steps_skipped, t=opt.scaler.adam_steps)), flush=True)
        compare(f'{which} hip-{prec} vs oracle64 (random init)', out, loss, G, r64[1], r64[0], r64[2], noise)
        sd = net.state_dict()
        perr = max(rel(sd[n], r64[3][n]) for n in P if n not in noise)
        print(json.dumps(dict(tag=f'{which} hip-{prec} params after Adam vs oracle64: worst rel-L2', v=perr)), flush=True)

    if os.environ.get('QUICK'):
        return
    # warm-up: 40 steps in HIP bf16 on fresh batches (fast), then compare one step at those weights
    net, opt = make(typ, cin, z, K, P, Bf, 'bf16')
    gg = torch.Generator().manual_seed(5)
    for s in range(int(os.environ.get('WARM', 40))):
        a = torch.rand(B, cin, size, size, generator=gg).cuda()
        b = torch.rand(B, cin, size, size, generator=gg).cuda()
        if which == 'c2':
            a, b = a * 2 - 1, b * 2 - 1
        l, _, _ = hip_step(net, opt, a, b)
    print(json.dumps(dict(tag='warm loss', v=float(l))), flush=True)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    Pw = {k: sd[k] for k in P}
    Bw = {k: sd[k] for k in Bf}
    w64 = oracle_step(plans, Pw, Bw, xs, xt, torch.float64)
    w32 = oracle_step(plans, Pw, Bw, xs, xt, torch.float32)
    compare(which + ' warm: oracle32 vs oracle64', w32[1], w32[0], w32[2], w64[1], w64[0], w64[2], noise)
    for prec in ('fp32', 'bf16'):
        net2, opt2 = make(typ, cin, z, K, Pw, Bw, prec)
        loss, out, G = hip_step(net2, opt2, xs, xt)
        compare(f'{which} warm: hip-{prec} vs oracle64', out, loss, G, w64[1], w64[0], w64[2], noise)

    # trajectories from the random init: bf16 HIP vs fp32 oracle, 20 steps on the same batch sequence
    net, opt = make(typ, cin, z, K, P, Bf, 'bf16')
    oP = {k: v.cuda().clone() for k, v in P.items()}
    oB = {k: v.cuda().clone() for k, v in Bf.items()}
    st = {}
    gg = torch.Generator().manual_seed(6)
    traj = []
    for s in range(20):
        a = torch.rand(B, cin, size, size, generator=gg).cuda()
        b = torch.rand(B, cin, size, size, generator=gg).cuda()
        if which == 'c2':
            a, b = a * 2 - 1, b * 2 - 1
        l, _, _ = hip_step(net, opt, a, b)
        ol, _, _ = O.transporter_train_step(plans, oP, oB, st, a, b)
        traj.append((float(l), float(ol)))
    print(json.dumps(dict(tag=which + ' trajectory (hip bf16, oracle fp32)', traj=traj)), flush=True)


if __name__ == '__main__':
    for w in sys.argv[1:]:
        main(w)
