"""Source + target frame of a Transporter pair through an encoder as ONE batch with per-half BatchNorm statistics (Unit.forward_pair,
the kp_*_pair entry points of include/kp_hip.h) against the two calls the reference makes (transporter.py:36-39: `with no_grad:
extract(xs)`, then `extract(xt)`).  Every pair launch runs each half exactly as the separate call would, so EVERYTHING must agree
bit for bit: both outputs, the running statistics after their two momentum updates, num_batches_tracked, and the gradients."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _unit(K, dt, seed=7):
    import torch.nn as nn
    from keypoints_amd.models import knn, vgg
    torch.manual_seed(seed)
    leaky = dict(nonlinearity=nn.LeakyReLU, nonlinearity_kwargs={'inplace': True})
    u = knn.Unit(1, K, vgg.make_layers(vgg.vgg_cfg['VGG_PONG_LAYERNECK'], **leaky)).cuda()
    for m in u.modules():   # non-trivial BatchNorm parameters and running statistics
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    u.precision = dt
    return u


def _state(u):
    return {k: v.detach().clone().cpu() for k, v in u.state_dict().items()}


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('K,out_t', [(4, False), (16, True), (16, False)])
@pytest.mark.parametrize('N,H,W', [(8, 84, 84), (16, 48, 64)])
def test_forward_pair_is_the_two_calls_bit_for_bit(dt, K, out_t, N, H, W):
    import keypoints_amd as ka
    from keypoints_amd.unit_engine import pair_ok
    g = torch.Generator().manual_seed(N + H)
    xs = torch.rand(N, 1, H, W, generator=g).cuda()
    xt = torch.rand(N, 1, H, W, generator=g).cuda()
    ua, ub = _unit(K, dt), _unit(K, dt)
    assert pair_ok(ua, N, 1, H, W, out_t)
    res = []
    for u, on in ((ua, False), (ub, True)):
        ka.set_pair_calls(on)
        try:
            assert pair_ok(u, N, 1, H, W, out_t) == on
            o_s, o_t = u.forward_pair(xs, xt, out_t=out_t)
            assert not o_s.requires_grad and o_t.requires_grad
            gout = torch.randn(o_t.shape, generator=torch.Generator().manual_seed(1)).to(o_t.dtype).cuda() * 0.1
            o_t.backward(gout)
            torch.cuda.synchronize()
            res.append((o_s.detach().float().cpu(), o_t.detach().float().cpu(), _state(u),
                        {n: p.grad.detach().clone().cpu() for n, p in u.named_parameters()}))
        finally:
            ka.set_pair_calls(True)
    (s0, t0, st0, g0), (s1, t1, st1, g1) = res
    assert torch.isfinite(s1).all() and torch.isfinite(t1).all()
    assert torch.equal(s0, s1) and torch.equal(t0, t1)
    for k in st0:
        assert torch.equal(st0[k], st1[k]), k          # weights untouched, running statistics after BOTH updates, num_batches_tracked
    assert any('num_batches_tracked' in k and int(st1[k]) == 2 for k in st1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
def test_transporter_step_with_pair_calls_is_bitwise_the_step_without(dt):
    import keypoints_amd as ka
    from keypoints_amd.models import transporter
    from keypoints_amd.optim import l2_reconstruction_loss
    B, H, W = 8, 84, 84
    g = torch.Generator().manual_seed(21)
    xs = torch.rand(B, 1, H, W, generator=g).cuda()
    xt = torch.rand(B, 1, H, W, generator=g).cuda()
    res = []
    for on in (False, True):
        ka.set_pair_calls(on)
        try:
            torch.manual_seed(5)
            net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 4, combine_mode='max').cuda().set_precision(dt)
            out = net(xs, xt)
            loss = l2_reconstruction_loss(out[0], xt)
            loss.backward()
            torch.cuda.synchronize()
            flat = [o for m in out for o in (m if isinstance(m, (tuple, list)) else (m,))]
            res.append(([o.detach().float().cpu() for o in flat], torch.cat([p.grad.flatten() for p in net.parameters()]).cpu(),
                        {k: v.detach().clone().cpu() for k, v in net.state_dict().items()}))
        finally:
            ka.set_pair_calls(True)
    (o0, g0, s0), (o1, g1, s1) = res
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    assert torch.equal(g0, g1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_pair_declines_where_it_has_no_kernels_and_the_two_calls_run():
    from keypoints_amd.unit_engine import pair_ok
    u = _unit(4, 'bf16')
    assert pair_ok(u, 8, 1, 84, 84)
    assert not pair_ok(u, 3, 1, 84, 84)                 # 3 x 7 056 pixels: not a multiple of the first layer's 64-pixel groups
    assert pair_ok(u, 4, 1, 84, 84)                     # (few statistics rows per half: the narrow finalize has the pair form at every row count)
    assert not pair_ok(u, 8, 1, 21, 21)                 # 441 pixels per image: not the kernels' granularity
    u.precision = 'fp32'
    assert not pair_ok(u, 8, 1, 84, 84)
    u.precision = 'bf16'
    u.eval()
    assert not pair_ok(u, 8, 1, 84, 84)                 # eval-mode BatchNorm: nothing to pair
    u.train()
    x = torch.rand(2, 1, 21, 21).cuda()
    o_s, o_t = u.forward_pair(x, x * 0.5)               # falls back to the two calls
    assert o_s.shape == o_t.shape == (2, 4, 21, 21) and not o_s.requires_grad and o_t.requires_grad


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# rel-L2 of a 16-bit Unit (three BatchNorm layers deep) against fp64.  Outputs and running statistics: the storage type's half-ulp
# (2^-12 fp16) per stored tensor, a handful of stored tensors in sequence -- measured 7e-4 / 1.2e-4 in fp16; bf16 = fp16 x 8 (3 fewer
# mantissa bits, the fixed rule of oracle.TOL16; measured 5.6e-3 / 7.5e-4).  Gradients along a random direction: a pre-activation
# inside the rounding error of zero takes the other LeakyReLU branch (derivative 1 <-> 0.01), a fraction f ~ 1e-3 (fp16) of the
# terms of every sum, and a sum of M random-sign terms with f M of them changed moves by sqrt(f) of its norm: measured 1.6-4.8e-2
# in fp16, 5.5-9.1e-2 in bf16 (x sqrt(8)).  A wrong or missing half would show as O(1).
PAIR_TOL = {'fp16': dict(out=2e-3, grad=0.1, stat=4e-4), 'bf16': dict(out=1.6e-2, grad=0.28, stat=3.2e-3)}


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('K,out_t', [(4, False), (16, True)])
def test_forward_pair_against_the_oracle(dt, K, out_t):
    """The pair launches against the CPU oracle directly (the bit-for-bit tests above compare them with the two HIP calls): the
    reference's two encoder calls of a Transporter step (transporter.py:36-39), source under no_grad then target, each with ITS OWN
    batch statistics, the running statistics updated twice in that order (nn.BatchNorm2d momentum 0.1), gradients of the target
    call only -- oracle.unit_forward in fp64 + autograd."""
    from keypoints_amd.unit_engine import pair_ok
    from oracle import kp_oracle as O
    N, H, W = 8, 84, 84
    g = torch.Generator().manual_seed(100 + K)
    xs = torch.rand(N, 1, H, W, generator=g) * 2 - 1
    xt = torch.rand(N, 1, H, W, generator=g) * 2 - 1
    u = _unit(K, dt)
    sd = _state(u)
    P = {'u.' + k: v.double().requires_grad_(True) for k, v in sd.items() if 'running' not in k and 'num_batches' not in k}
    Bf = {'u.' + k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items() if 'running' in k or 'num_batches' in k}
    plan = O.unit_plan(O.ENCODER_CORES['VGG_PONG_LAYERNECK'], 1, K, O.LEAKY)
    assert pair_ok(u, N, 1, H, W, out_t)
    o_s, o_t = u.forward_pair(xs.cuda(), xt.cuda(), out_t=out_t)
    wgt = torch.randn(N, K, H, W, generator=g) * 0.1
    if out_t:   # T [N][H][W][K]
        gout = wgt.permute(0, 2, 3, 1).contiguous().to(o_t.dtype)
        o_t.backward(gout.cuda())
        wgt = gout.float().permute(0, 3, 1, 2)
        o_s, o_t = o_s.float().permute(0, 3, 1, 2), o_t.float().permute(0, 3, 1, 2)
    else:
        o_t.backward(wgt.cuda())
    with torch.no_grad():
        oo_s = O.unit_forward(plan, P, 'u', xs.double(), Bf, training=True)
    oo_t = O.unit_forward(plan, P, 'u', xt.double(), Bf, training=True)
    (oo_t * wgt.double()).sum().backward()
    tol = PAIR_TOL[dt]
    assert _rel(o_s, oo_s) <= tol['out'] and _rel(o_t, oo_t) <= tol['out'], (_rel(o_s, oo_s), _rel(o_t, oo_t))
    after = _state(u)
    for k, v in Bf.items():
        if 'num_batches' in k:
            assert int(after[k[2:]]) == int(v) == 2
        else:
            assert _rel(after[k[2:]], v) <= tol['stat'], (k, _rel(after[k[2:]], v))
    worst = {}
    for n, p in u.named_parameters():
        og = P['u.' + n].grad
        if og is None or float(og.norm()) < 1e-9 * og.numel():   # conv biases in front of a BatchNorm: analytically zero
            continue
        worst[n] = _rel(p.grad, og)
    assert len(worst) >= 6 and max(worst.values()) <= tol['grad'], worst

