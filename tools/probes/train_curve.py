"""Training parity over many steps (BASELINE config 1 shape: Pong-grey 32x32, K=3, 32 pairs -- the reference's own CPU-runnable
case, configs/transporter_pong_grey.yaml): the CPU oracle (oracle/kp_oracle.py transporter_train_step, fp32) and the HIP path in
strict fp32, fp16 (+ loss scaling) and bf16 take the SAME initial weights and the SAME stream of structured pairs through STEPS Adam
steps (lr 1e-3).  Trajectories of a non-convex problem separate (Adam amplifies last bits), so the comparison is the training curve:
the loss at marked steps and its mean over the last 50.        python tools/probes/train_curve.py [steps]"""
import sys
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import keypoints_amd as ka
from keypoints_amd import amp
from keypoints_amd.models import transporter
from oracle import kp_oracle as O
from structured import structured_pairs

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
LR, B, S = 1e-3, 32, 32
MARKS = [0, 10, 30, 100, 200, 400, STEPS - 1]
plans = O.transporter_plans('VGG_PONG_LAYERNECK', 1, 16, 3)
P0, B0 = O.init_params(plans, seed=5)


def batches():
    g = torch.Generator().manual_seed(11)
    for _ in range(STEPS):
        yield structured_pairs(B, S, g, channels=1)


def run_oracle():
    P = {k: v.clone() for k, v in P0.items()}; Bf = {k: v.clone() for k, v in B0.items()}
    state, out = {}, []
    torch.set_num_threads(16)
    for xs, xt in batches():
        loss, _, _ = O.transporter_train_step(plans, P, Bf, state, xs, xt, lr=LR)
        out.append(float(loss))
    return out


def run_hip(precision):
    net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 3)
    net.load_state_dict({**P0, **B0}, strict=True)
    net = net.cuda().set_precision(precision)
    opt = ka.FusedAdam(net.parameters(), lr=LR).bind(net)
    if precision == 'fp16':
        net, opt = amp.initialize(net, opt, opt_level='O2')
    out = []
    for xs, xt in batches():
        xs, xt = xs.cuda(), xt.cuda()
        opt.zero_grad()
        loss = ka.l2_reconstruction_loss(net(xs, xt)[0], xt)
        if precision == 'fp16':
            with amp.scale_loss(loss, opt) as sl:
                sl.backward()
        else:
            loss.backward()
        opt.step()
        out.append(float(loss))
    return out


curves = {'oracle fp32 (CPU)': run_oracle()}
for p in ('fp32', 'fp16', 'bf16'):
    curves['HIP ' + p] = run_hip(p)
print(f'config 1 shape, {B} pairs, {STEPS} Adam steps, lr {LR}: loss at step ' + ' / '.join(str(m) for m in MARKS) + ' | mean of the last 50')
ref = curves['oracle fp32 (CPU)']
for name, c in curves.items():
    tail = sum(c[-50:]) / 50
    worst = max(abs(a - b) / b for a, b in zip(c, ref))
    print(f'{name:20s} ' + ' / '.join(f'{c[m]:.4e}' for m in MARKS) + f' | {tail:.4e}   worst step vs oracle {100 * worst:.1f} %, tail vs oracle {100 * (tail / (sum(ref[-50:]) / 50) - 1):+.1f} %')
