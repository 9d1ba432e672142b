#!/usr/bin/env python3
"""How far are the 16-bit HIP modes from the fp64 oracle on model F once the network is CONDITIONED (trained for a while on
structured pairs) instead of at random init?  Trains with the strict-fp32 HIP path, hands the weights to the oracle and to
fresh bf16 / fp16 HIP networks, compares one train step at 128 pairs (x_t <= 2e-2 max|x_t|, keypoints 5e-3)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch

import keypoints_amd as ka
from keypoints_amd.models import transporter
from oracle import kp_oracle as O
from structured import structured_pairs


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    steps_list = [int(v) for v in (sys.argv[1:] or ['0', '50', '200', '600'])]
    train_prec = os.environ.get('COND_PREC', 'fp32')
    tb = int(os.environ.get('COND_BATCH', '32'))
    plans = O.transporter_plans('F', 3, 64, 10)
    P, Bf = O.init_params(plans, seed=3)
    net = transporter.make('F', 3, 64, 10)
    net.load_state_dict({**P, **Bf}, strict=True)
    net = net.cuda().set_precision(train_prec)
    opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
    g = torch.Generator().manual_seed(99)
    gt = torch.Generator().manual_seed(7)
    xs_e, xt_e = structured_pairs(128, 128, gt, device='cuda')
    done = 0
    loss = float('nan')
    for target in steps_list:
        t0 = time.time()
        while done < target:
            a, b = structured_pairs(tb, 128, g, device='cuda')
            opt.zero_grad()
            loss = ka.l2_reconstruction_loss(net(a, b)[0], b)
            loss.backward()
            opt.step()
            done += 1
        torch.cuda.synchronize()
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        Pn = {k: sd[k] for k in P}
        Bn = {k: sd[k] for k in Bf}
        oP = {k: v.double().clone() for k, v in Pn.items()}
        oB = {k: (v.double() if v.dtype.is_floating_point else v).clone() for k, v in Bn.items()}
        with torch.backends.cudnn.flags(enabled=False):
            l64, o64, G64 = O.transporter_train_step(plans, oP, oB, {}, xs_e.double(), xt_e.double())
        line = [f'steps {done:4d} ({time.time() - t0:.0f}s) train-loss {float(loss):.5f} eval-loss {float(l64):.5f} max|x_t| {float(o64[0].abs().max()):.3f}']
        for prec in ('fp32', 'bf16', 'fp16'):
            n2 = transporter.make('F', 3, 64, 10)
            n2.load_state_dict({**Pn, **Bn}, strict=True)
            n2 = n2.cuda().set_precision(prec)
            o2 = ka.FusedAdam(n2.parameters(), lr=1e-4).bind(n2)
            if prec == 'fp16':
                o2.scaler = ka.amp.LossScaler('cuda')
            o2.zero_grad()
            out = n2(xs_e, xt_e)
            l = ka.l2_reconstruction_loss(out[0], xt_e)
            with ka.amp.scale_loss(l, o2) as sl:
                sl.backward()
            torch.cuda.synchronize()
            xm = float((out[0].double() - o64[0]).abs().max()) / float(o64[0].abs().max())
            km = float((out[2].double() - o64[2]).abs().max())
            line.append(f'  {prec}: x_t {xm:.2e}*max rel {rel(out[0], o64[0]):.2e}  k {km:.2e}  phi {rel(out[1], o64[1]):.2e} '
                        f'loss {abs(float(l) - float(l64)) / float(l64):.1e}')
            del n2, o2, out
        print('\n'.join(line), flush=True)


if __name__ == '__main__':
    main()
