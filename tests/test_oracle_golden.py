"""CPU: the oracle (oracle/kp_oracle.py) against the golden vectors that tools/make_golden.py captured
from the reference itself.  This is what pins the oracle; the GPU tests then compare HIP vs oracle/golden."""
import os

import numpy as np
import pytest
import torch

from oracle import kp_oracle as O

BOTTLENECK = ['c4', 'c5', 'c1', 'pong', 'ragged', 'centre5', 'corner5', 'spike16', 'uniform', 'pm80', 'one_pixel']


def close(a, b, tol=1e-6):
    a = a.detach().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize('tag', BOTTLENECK)
def test_bottleneck(golden, tag):
    z = golden('bottleneck.npz')
    h = torch.tensor(z[f'{tag}.h'], requires_grad=True)
    k, (p, q) = O.spatial_logsoftmax(h)
    close(k, z[f'{tag}.k']); close(p, z[f'{tag}.p']); close(q, z[f'{tag}.q'])
    (k * torch.tensor(z[f'{tag}.wk'])).sum().backward()
    close(h.grad, z[f'{tag}.dh'])
    close(O.spatial_softmax(h.detach())[0], z[f'{tag}.k_soft'])
    kd = torch.tensor(z[f'{tag}.k'], requires_grad=True)
    H, W = z[f'{tag}.m'].shape[2:]
    m = O.gaussian_like(kd, H, W)
    close(m, z[f'{tag}.m'])
    (m * torch.tensor(z[f'{tag}.wm'])).sum().backward()
    close(kd.grad, z[f'{tag}.dk'], 1e-5)
    close(O.point_map(kd.detach(), H, W), z[f'{tag}.point_map'])


def test_spike_and_uniform_known_answers(golden):
    """tests/tests.py:237-241 spike -> (0.4327, 0.5673) [SURVEY 4]; uniform map -> (0.5, 0.5)."""
    h = torch.zeros(1, 1, 16, 16); h[0, 0, 0, 15] = 20.0
    k = O.spatial_logsoftmax(h)[0][0, 0]
    assert abs(k[0].item() - 0.4327) < 1e-4 and abs(k[1].item() - 0.5673) < 1e-4
    assert (O.spatial_logsoftmax(torch.zeros(1, 2, 9, 7))[0] - 0.5).abs().max() < 1e-6
    peak = O.gaussian_like(torch.tensor([[[5 / 15, 10 / 15]]]), 16, 16).max().item()
    assert abs(peak - np.exp(-np.sqrt(1e-6) * 50)) < 1e-6   # exp(-0.05) = 0.9512


@pytest.mark.parametrize('mode', ['max', 'sum_and_clamp', 'loop', 'pretrained_network'])
def test_transport(golden, mode):
    z = golden('transport.npz')
    t = [torch.tensor(z[n]) for n in ('phi_s', 'phi_t', 'm_s', 'm_t')]
    ms, mt = (t[2], t[3]) if mode != 'pretrained_network' else (t[2][:, :1], t[3][:, :1])
    phi, mask_s, mask_t = O.transport(t[0], t[1], ms, mt, mode)
    close(phi, z[f'{mode}.phi']); close(mask_s, z[f'{mode}.mask_s']); close(mask_t, z[f'{mode}.mask_t'])


@pytest.mark.parametrize('tag,cin,cout,tail,slope', [('in3', 3, 64, None, 0.01), ('c64', 32, 64, 'M', 0.01),
                                                      ('dec', 32, 16, 'U', 0.0), ('odd', 5, 7, None, 0.01)])
def test_layer_block(golden, tag, cin, cout, tail, slope):
    z = golden('layers.npz')
    plan = [dict(op='conv3', key='b.1', bn='b.2', cin=cin, cout=cout, slope=slope)]
    if tail == 'M':
        plan.append(dict(op='pool'))
    if tail == 'U':
        plan.append(dict(op='up'))
    P = {'n.b.1.weight': torch.tensor(z[f'{tag}.weight'], requires_grad=True), 'n.b.1.bias': torch.tensor(z[f'{tag}.bias']),
         'n.b.2.weight': torch.tensor(z[f'{tag}.gamma'], requires_grad=True),
         'n.b.2.bias': torch.tensor(z[f'{tag}.beta'], requires_grad=True)}
    Bf = {'n.b.2.running_mean': torch.zeros(cout), 'n.b.2.running_var': torch.ones(cout)}
    x = torch.tensor(z[f'{tag}.x'], requires_grad=True)
    h = O.unit_forward(plan, P, 'n', x, Bf)
    close(h, z[f'{tag}.y'])
    (h * torch.tensor(z[f'{tag}.w'])).sum().backward()
    close(x.grad, z[f'{tag}.dx'], 1e-5)
    close(P['n.b.1.weight'].grad, z[f'{tag}.dweight'], 1e-5)
    close(P['n.b.2.weight'].grad, z[f'{tag}.dgamma'], 1e-5)
    close(Bf['n.b.2.running_var'], z[f'{tag}.running_var'])


def test_unit_forward_accepts_a_plan_without_a_head():
    """A plan that ends in a conv3 / pool / up step (no 1x1 head behind it) runs: tools/make_golden.py feeds single
    blocks (the round-1 oracle peeked past the end of such a plan and the committed generator crashed)."""
    plan = [dict(op='conv3', key='b.1', bn='b.2', cin=3, cout=4, slope=0.01)]
    P = {'n.b.1.weight': torch.randn(4, 3, 3, 3), 'n.b.1.bias': torch.zeros(4), 'n.b.2.weight': torch.ones(4),
         'n.b.2.bias': torch.zeros(4)}
    assert O.unit_forward(plan, P, 'n', torch.randn(2, 3, 6, 6)).shape == (2, 4, 6, 6)
    assert O.unit_forward(plan + [dict(op='pool')], P, 'n', torch.randn(2, 3, 6, 6)).shape == (2, 4, 3, 3)


@pytest.mark.skipif(not os.path.isdir('/root/reference/keypoints'), reason='the reference exists in the build container only')
def test_fixture_generator_check_mode_small():
    """tools/make_golden.py --check regenerates the fixtures from the imported reference into a temp dir and demands
    bit-equality with tests/golden/ (small fixtures here; the full run's log is tests/golden/CHECK.log)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'make_golden.py'), '--check', 'bottleneck', 'transport',
                        'layers', 'adam', 'pong32', 'tps'], capture_output=True, text=True, timeout=600,
                       env={**os.environ, 'PYTHONDONTWRITEBYTECODE': '1'})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_adam(golden):
    z = golden('adam.npz')
    P = {'p': torch.tensor(z['p0'])}
    st = {}
    for t in range(1, 4):
        O.adam_update(P, {'p': torch.tensor(z[f'g{t}'])}, st)
        close(P['p'], z[f'p{t}'], 1e-7)


@pytest.mark.parametrize('tag,size,K', [('32', 32, 3), ('84', 84, 4)])
def test_transporter_pong_train_steps(golden, tag, size, K):
    z = golden(f'transporter_pong_{tag}.npz')
    plans = O.transporter_plans('VGG_PONG_LAYERNECK', 1, 16, K)
    P = {k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith('p0.')}
    Bf = {k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith('b0.')}
    xs, xt = torch.tensor(z['xs']), torch.tensor(z['xt'])
    noise = {f"{n}.{s['key']}.bias" for n, pl in plans.items() for s in pl if s['op'] == 'conv3'}
    st = {}
    for step in range(1, 4):
        loss, out, G = O.transporter_train_step(plans, P, Bf, st, xs, xt)
        close(loss, z[f'loss{step}'])
        if step == 1:
            close(out[0], z['x_t'], 1e-5); close(out[2], z['k']); close(out[3], z['m']); close(out[1], z['phi'], 1e-5)
            close(out[6], z['mask_s']); close(out[7], z['mask_t'])
            for n in P:
                if n not in noise:
                    close(G[n], z['g.' + n], 1e-5)
    for n in P:
        if n not in noise:
            close(P[n], z['p3.' + n])


def test_transporter_F_forward(golden):
    """Model F 128x128 K=10 forward + loss (the north-star architecture) against the reference's outputs."""
    z = golden('transporter_F_128.npz')
    plans = O.transporter_plans('F', 3, 64, 10)
    P, Bf = O.init_params(plans, seed=int(z['seed_params']))
    assert abs(sum(v.double().abs().sum().item() for v in P.values()) - float(z['checksum'])) < 1e-6 * float(z['checksum'])
    g = torch.Generator().manual_seed(int(z['seed_inputs']))
    xs = torch.rand(2, 3, 128, 128, generator=g); xt = torch.rand(2, 3, 128, 128, generator=g)
    with torch.no_grad():
        out = O.transporter_forward(plans, P, xs, xt, Bf)
    close(out[0], z['x_t'], 1e-5); close(out[2], z['k']); close(out[3], z['m'])
    close(O.l2_reconstruction_loss(out[0], xt), z['loss'])
    assert abs(O.conv_flops_per_pair(plans, 128, 128) / 1e9 - 235.23) < 0.01   # BASELINE.md work accounting


NOBN = [('enc', 3, 5, 'VGG_PONG', True), ('dec', 16, 2, 'VGG_PONG', False), ('pac', 1, 4, 'VGG_PACMAN', True)]


@pytest.mark.parametrize('tag,cin,cout,typ,enc', NOBN)
def test_unit_without_batchnorm(golden, tag, cin, cout, typ, enc):
    """`Unit(..., batch_norm=False)` over `make_layers(cfg, batch_norm=False)` (knn.py:111-125, vgg.py:16-38) as the reference
    computed it (tools/make_golden.py make_nobn): forward, input gradient, every parameter gradient."""
    z = golden('nobn.npz')
    cfg = (O.ENCODER_CORES if enc or tag == 'pac' else O.DECODER_CORES)[typ]
    plan = O.unit_plan(cfg, cin, cout, O.LEAKY if enc else 0.0, batch_norm=False)
    assert all(s.get('bn') is None for s in plan if s['op'] == 'conv3')
    P = {n: torch.tensor(z[f'{tag}.p.{n[2:]}'], requires_grad=True) for n in O.param_shapes({'n': plan})}
    x = torch.tensor(z[f'{tag}.x'], requires_grad=True)
    y = O.unit_forward(plan, P, 'n', x, None, True)
    close(y, z[f'{tag}.y'])
    (y * torch.tensor(z[f'{tag}.w'])).sum().backward()
    close(x.grad, z[f'{tag}.dx'])
    for n, p in P.items():
        close(p.grad, z[f'{tag}.g.{n[2:]}'])
