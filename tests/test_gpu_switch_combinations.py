"""The engine's switches select between EQUIVALENT paths (bit-identical, or equal up to the order of fp32 / fp64
sums).  The per-switch tests toggle one at a time; this one draws COMBINATIONS ("tests cover defaults and single
toggles, not combinations"): every drawn setting of the nine runtime switches and three library knobs runs the same two train steps of
a Pong-class net and of a small model-F net, and must land where the default setting lands."""
import itertools
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

KP_CFG_BRED_PATCH, KP_CFG_FIN_NT, KP_CFG_FIRST_ROWS = 9, 17, 14


def _settings(n, seed):
    rng = random.Random(seed)
    names = ('multi_stream', 'fuse_bn_input', 'fuse_bn_reduce', 'fuse_first_layer', 'fold_finalize', 'pair_calls', 't_interchange',
             'lazy_outputs', 'keep_saved')
    out = [dict.fromkeys(names, None)]        # None = leave at the default
    defaults = dict(multi_stream=True, fuse_bn_input=True, fuse_bn_reduce=True, fuse_first_layer=True, fold_finalize=False, pair_calls=True,
                    t_interchange=True, lazy_outputs=True, keep_saved=False)
    for _ in range(n):
        s = {k: (not defaults[k]) if rng.random() < 0.45 else defaults[k] for k in names}
        s['knobs'] = {KP_CFG_BRED_PATCH: rng.choice((0, 1, 2)), KP_CFG_FIN_NT: rng.choice((256, 1024)), KP_CFG_FIRST_ROWS: rng.choice((0, 1))}
        out.append(s)
    return out, defaults


def _apply(s, defaults):
    import keypoints_amd as ka
    from keypoints_amd import runtime
    from keypoints_amd._lib import call
    v = lambda k: defaults[k] if s.get(k) is None else s[k]
    ka.set_multi_stream(v('multi_stream')); ka.set_fuse_bn_input(v('fuse_bn_input')); runtime.set_fuse_bn_reduce(v('fuse_bn_reduce'))
    ka.set_fuse_first_layer(v('fuse_first_layer')); ka.set_fold_finalize(v('fold_finalize')); ka.set_pair_calls(v('pair_calls'))
    ka.set_t_interchange(v('t_interchange')); ka.set_lazy_outputs(v('lazy_outputs')); runtime.set_keep_saved_activations(v('keep_saved'))
    knobs = s.get('knobs') or {KP_CFG_BRED_PATCH: 1, KP_CFG_FIN_NT: 256, KP_CFG_FIRST_ROWS: 1}
    for k, val in knobs.items():
        call('kp_config', k, val)


@pytest.mark.parametrize('model,precision', [('VGG_PONG_LAYERNECK', 'bf16'), ('F', 'fp16'), ('F', 'fp32')])
def test_switch_combinations_land_where_the_default_lands(model, precision):
    import keypoints_amd as ka
    from keypoints_amd.models import transporter
    cin, z, K, size, B = (1, 16, 4, 84, 16) if model != 'F' else (3, 64, 10, 64, 8)
    g = torch.Generator().manual_seed(3)
    xs = (torch.rand(B, cin, size, size, generator=g) * 2 - 1).cuda()
    xt = (torch.rand(B, cin, size, size, generator=g) * 2 - 1).cuda()
    settings, defaults = _settings(7, seed=sum(map(ord, model + precision)))   # (a FIXED draw per case: str hashes change per process)
    results = []
    try:
        for s in settings:
            _apply(s, defaults)
            torch.manual_seed(11)
            net = transporter.make(model, cin, z, K).cuda().set_precision(precision)
            opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
            if precision == 'fp16':
                opt.scaler = ka.amp.LossScaler('cuda')
            losses = []
            for _ in range(2):
                opt.zero_grad()
                out = net(xs, xt)
                loss = ka.l2_reconstruction_loss(out[0], xt)
                with ka.amp.scale_loss(loss, opt) as scaled:
                    scaled.backward()
                opt.step()
                losses.append(float(loss.detach()))
            torch.cuda.synchronize()
            phi = out[1].float() + 0      # (a LazyNCHW materialises here)
            results.append((losses, out[0].detach().float().clone(), out[2].detach().clone(), phi.detach().clone(),
                            torch.cat([p.detach().flatten() for p in net.parameters()]).clone(), s))
    finally:
        _apply(settings[0], defaults)
    l0, x0, k0, phi0, p0, _ = results[0]
    # 16-bit storage at random init: a statistic's last bit flips single roundings that 25 BatchNorm layers amplify (the same bounds as
    # tests/test_gpu_streams.py / test_gpu_dist_nccl.py use for summation-order-equivalent paths); strict fp32: summation order only
    tol_x, tol_k, tol_l = (2e-5, 2e-6, 1e-6) if precision == 'fp32' else (3e-2, 4e-3, 2e-3)
    if model != 'F' and precision != 'fp32':
        tol_x, tol_k, tol_l = 2e-2, 2e-3, 1e-3       # (the T route off rounds the head's output once less: INTEGRATION.md, deviations)
    for losses, x, k, phi, p, s in results[1:]:
        assert all(abs(a - b) <= tol_l * abs(b) for a, b in zip(losses, l0)), (s, losses, l0)
        assert float((x - x0).norm() / x0.norm()) <= tol_x, s
        assert float((k - k0).abs().max()) <= tol_k, s
        assert float((phi - phi0).norm() / phi0.norm()) <= tol_x, s
        # two Adam steps of lr 1e-4: a parameter moves by <= 1e-4 per step, so two runs are <= 4e-4 apart at the few parameters whose
        # (near-zero) gradient changes sign under the storage rounding; the bulk agrees far more closely
        assert float((p - p0).abs().max()) <= 4.1e-4, s
        assert float((p - p0).abs().mean()) <= (1e-7 if precision == 'fp32' else 2e-5), s
