"""GPU diagnostic: error of the HIP Transporter F path vs goldens / oracle variants (prints, no asserts)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import keypoints_amd as ka
from keypoints_amd.models import transporter
from oracle import kp_oracle as O

z = np.load(os.path.join(ROOT, 'tests/golden/transporter_F_128.npz'))
plans = O.transporter_plans('F', 3, 64, 10)
P, Bf = O.init_params(plans, seed=int(z['seed_params']))
g = torch.Generator().manual_seed(int(z['seed_inputs']))
xs = torch.rand(2, 3, 128, 128, generator=g); xt = torch.rand(2, 3, 128, 128, generator=g)


def run_hip(prec):
    net = transporter.make('F', 3, 64, 10)
    net.load_state_dict({**P, **Bf}, strict=True)
    net = net.cuda().set_precision(prec)
    out = net(xs.cuda(), xt.cuda())
    loss = ka.l2_reconstruction_loss(out[0], xt.cuda())
    loss.backward()
    torch.cuda.synchronize()
    return net, out, loss


for prec in ('fp32', 'bf16'):
    net, out, loss = run_hip(prec)
    print(f'== {prec}: loss {loss.item():.6f} ref {float(z["loss"]):.6f}')
    print('  x_t err vs ref32', (out[0].cpu() - torch.tensor(z['x_t'])).abs().max().item(), 'scale', np.abs(z['x_t']).max(),
          'ref noise', float(z['x_t_noise']))
    print('  k err', (out[2].cpu() - torch.tensor(z['k'])).abs().max().item(), ' heat err',
          (out[5].cpu() - torch.tensor(z['heatmap'])).abs().max().item(), 'scale', np.abs(z['heatmap']).max())
    worst = []
    for n, p in net.named_parameters():
        gn64 = float(z['g64norm.' + n])
        if gn64 < 1e-8:
            continue
        idx = torch.tensor(z['gidx.' + n]).cuda()
        got = p.grad.flatten()[idx].cpu().double().numpy()
        e_s = np.linalg.norm(got - z['g64val.' + n]) / (np.linalg.norm(z['g64val.' + n]) + 1e-30)
        e_n = abs(float(p.grad.norm()) - gn64) / gn64
        worst.append((e_s / max(1e-9, float(z['gnoise.' + n])), e_s, e_n, float(z['gnoise.' + n]), n))
    worst.sort()
    for w in worst[-5:]:
        print('  grad  ratio-to-ref-noise %.2f  sample-relL2 %.2e  norm-rel %.2e  refnoise %.2e  %s' % w)
    if prec == 'bf16':
        t = time.time()
        oP = {k: v.clone() for k, v in P.items()}; oB = {k: v.clone() for k, v in Bf.items()}
        ol, oo, oG = O.transporter_train_step(plans, oP, oB, {}, xs, xt, quant=O.bf16_storage)
        print('  oracle-bf16-emulation took', time.time() - t)
        print('  x_t err vs emu', (out[0].cpu() - oo[0]).abs().max().item(), ' emu vs ref32', (oo[0] - torch.tensor(z['x_t'])).abs().max().item())
        print('  k err vs emu', (out[2].cpu() - oo[2]).abs().max().item(), 'emu vs ref', (oo[2] - torch.tensor(z['k'])).abs().max().item())
        print('  loss emu', ol.item())
        ref = torch.tensor(z['x_t'])
        rl = lambda a, b: ((a - b).norm() / b.norm()).item()
        print('  x_t relL2: hip-vs-ref %.3e  emu-vs-ref %.3e  hip-vs-emu %.3e' % (rl(out[0].cpu(), ref), rl(oo[0], ref), rl(out[0].cpu(), oo[0])))
        rows2 = []
        for n, p in net.named_parameters():
            gn64 = float(z['g64norm.' + n])
            if gn64 < 1e-8: continue
            idx = torch.tensor(z['gidx.' + n])
            e_h = np.linalg.norm(p.grad.flatten().cpu()[idx].double().numpy() - z['g64val.' + n]) / np.linalg.norm(z['g64val.' + n])
            e_e = np.linalg.norm(oG[n].flatten()[idx].double().numpy() - z['g64val.' + n]) / np.linalg.norm(z['g64val.' + n])
            rows2.append((e_h / max(e_e, 1e-9), e_h, e_e, n))
        rows2.sort()
        print('  grad sample err vs fp64: hip / emu ratios: median %.2f' % rows2[len(rows2)//2][0])
        for r in rows2[-6:]: print('    ratio %.2f hip %.3e emu %.3e %s' % r)
        hs = sorted(r[1] for r in rows2); es = sorted(r[2] for r in rows2)
        print('  hip err quantiles', hs[len(hs)//2], hs[-1], ' emu', es[len(es)//2], es[-1])
        rows = []
        for n, p in net.named_parameters():
            gn = oG[n].norm().item()
            if gn < 1e-8: continue
            rows.append(((p.grad.cpu() - oG[n]).norm().item() / gn, n))
        rows.sort()
        print('  grad rel-L2 vs emu: median %.3e worst %s' % (rows[len(rows)//2][0], rows[-3:]))
