"""One complete train step (forward, loss, every gradient, Adam, BatchNorm running statistics) at the BASELINE batch
sizes, HIP path against the oracle.

The checker is oracle/kp_oracle.py run ON THE GPU (it is plain torch and device-agnostic) with MIOpen switched off, so
that torch's native im2col + rocBLAS GEMM convolution does the arithmetic: exact fp32 / fp64 (no Winograd / FFT
algorithms), and an fp64 run of 128 pairs at 128x128 takes 4 s.  The fp64 run is the yardstick; the oracle's own fp32
run against it gives the reference's noise floor on the very same input, which is how the gradient tolerances are set
("about 10x the floor" for outputs; gradients through 25 BatchNorm layers at random init are noisy in
the reference itself: 8e-3 median per-tensor rel-L2 between ITS fp32 and fp64 runs at 128 pairs).

Calibration runs behind the numbers below: tools/probes/fullsize_parity2.py (MI355X, this round):
  config 4, 128 pairs, fp32 strict:  x_t 1.1e-4 max|x_t|, k 6.4e-6, m 1.5e-4, loss 1e-7, gradients = the oracle's own
                                     fp32 noise (median 8.2e-3 vs 7.7e-3)
  config 4, 128 pairs, bf16:         x_t 2.1e-1 rel-L2, k 2.6e-2, loss 3.5e-4   -- see test_c4_bf16 below
  config 2, 256 pairs, bf16:         x_t 1.6e-2 max|x_t|, k 1.2e-4, loss 8e-5   -- inside SURVEY's 16-bit tolerance
  config 4, 128 pairs, fp16:         x_t 3.0e-2 rel-L2 (5.2e-2 max|x_t|), k 3.5e-3, loss 1.1e-4, phi 5.6e-3
  config 2, 256 pairs, fp16:         x_t 2.1e-3 max|x_t|, k 1.6e-5, loss 1.3e-5
"""
import contextlib
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # tests/structured.py

LR = 1e-4


@contextlib.contextmanager
def native_conv():
    """MIOpen off for the oracle: exact-arithmetic GEMM convolution, fp64 capable, no kernel search."""
    with torch.backends.cudnn.flags(enabled=False):
        yield


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def maxerr(a, b):
    return float((a.double() - b.double()).abs().max())


def noise_names(plans):
    """conv biases in front of train-mode BatchNorm: analytically zero gradient (pure rounding noise in the reference)."""
    return {f"{n}.{s['key']}.bias" for n, pl in plans.items() for s in pl if s['op'] == 'conv3'}


def oracle_step(O, step_fn, plans, P, Bf, dtype, *inputs, **kw):
    oP = {k: v.to('cuda', dtype).clone() for k, v in P.items()}
    oB = {k: (v.to('cuda', dtype) if v.dtype.is_floating_point else v.to('cuda')).clone() for k, v in Bf.items()}
    ins = [t.to(dtype) if t is not None else None for t in inputs]
    with native_conv():
        loss, out, G = step_fn(plans, oP, oB, {}, *ins, **kw)
    return loss, out, G, oP, oB


def hip_net(make_fn, P, Bf, precision, **scaler_kw):
    import keypoints_amd as ka
    net = make_fn()
    net.load_state_dict({**P, **Bf}, strict=True)
    net = net.cuda().set_precision(precision)
    opt = ka.FusedAdam(net.parameters(), lr=LR).bind(net)
    if precision == 'fp16':   # the reference's O2 recipe: fp16 storage + dynamic loss scaling (transporter.py:51,84-86)
        opt.scaler = ka.amp.LossScaler('cuda', **scaler_kw)
    return net, opt


def check_fp32_step(net, out, loss, G, P0, r64, r32, noise):
    """HIP strict-fp32 step against the fp64 oracle, with the fp32 oracle's own distance to it as the noise floor."""
    l64, o64, G64, P64, B64 = r64
    l32, o32, G32, _, _ = r32
    x_t, k, m = out
    scale = float(o64[0].abs().max())
    assert maxerr(x_t, o64[0]) <= 1.5e-4 * scale
    assert maxerr(k, o64[2]) <= 2e-5 and maxerr(m, o64[3]) <= 2.5e-4
    assert abs(float(loss) - float(l64)) <= 1e-5 * float(l64)
    ratios = []
    for n, g64 in G64.items():
        if n in noise:
            assert float(G[n].abs().max()) == 0.0, n                      # exactly zero here; rounding noise in the reference
            continue
        e_hip, e_ref = rel(G[n], g64), rel(G32[n], g64)
        assert e_hip <= 2.5 * e_ref + 2e-3, (n, e_hip, e_ref)      # per tensor (some have 10 entries): loose; the median below is the aggregate
        ratios.append(e_hip / max(e_ref, 1e-6))
    assert np.median(ratios) <= 1.25, np.median(ratios)
    # Adam, step 1: every parameter moves by lr * g / (|g| + eps) ~ lr * sign(g); a sign can only differ where the
    # gradient is at its noise level
    sd = net.state_dict()
    for n, p64 in P64.items():
        if n in noise:
            continue
        d_hip = sd[n].double() - P0[n].cuda().double()
        d_ref = p64 - P0[n].cuda().double()
        assert float(d_hip.abs().max()) <= 1.01 * LR               # (+ an ulp of the parameter itself)
        assert float((d_hip - d_ref).abs().mean()) <= 0.05 * LR, (n, float((d_hip - d_ref).abs().mean()) / LR)
    for n, b64 in B64.items():
        if n.endswith('num_batches_tracked'):
            assert int(sd[n]) == int(b64)
        else:
            assert maxerr(sd[n], b64) <= 1e-5 * max(1.0, float(b64.abs().max())), n


def hip_step(net, opt, loss_fn):
    import keypoints_amd as ka
    opt.zero_grad()
    loss, out = loss_fn()
    with ka.amp.scale_loss(loss, opt) as scaled:        # identity without a scaler (fp32 / bf16)
        scaled.backward()
    inv = 1.0 / opt.scaler.scale if opt.scaler is not None else 1.0
    G = {n: p.grad.detach().clone() * inv for n, p in net.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    return loss.detach(), out, G


def _transporter_case(typ, cin, z, K, size, B, pm1):
    from keypoints_amd.models import transporter
    from oracle import kp_oracle as O
    plans = O.transporter_plans(typ, cin, z, K)
    P, Bf = O.init_params(plans, seed=3)
    g = torch.Generator().manual_seed(21)
    xs = torch.rand(B, cin, size, size, generator=g)
    xt = torch.rand(B, cin, size, size, generator=g)
    if pm1:   # Pong frames are Normalize(.5, .5)-ed (keypoints/ds/datasets.py:292-295)
        xs, xt = xs * 2 - 1, xt * 2 - 1
    return O, plans, P, Bf, xs.cuda(), xt.cuda(), (lambda: transporter.make(typ, cin, z, K))


def _run_transporter(net, opt, xs, xt):
    import keypoints_amd as ka

    def f():
        out = net(xs, xt)
        return ka.l2_reconstruction_loss(out[0], xt), out
    loss, out, G = hip_step(net, opt, f)
    return loss, (out[0], out[2], out[3]), G, out


@pytest.mark.parametrize('cfg', ['c4', 'c2', 'c5'])
def test_train_step_fp32_full_batch(cfg):
    """Strict-fp32 mode, one whole train step at the BASELINE batch: config 4 (F 128x128 K=10, 128 pairs), config 2
    (Pong-grey 84x84 K=4, 256 pairs), config 5 shape (F 256x256 K=30; 16 pairs: the strict mode runs VALU kernels)."""
    typ, cin, z, K, size, B, pm1 = dict(c4=('F', 3, 64, 10, 128, 128, False), c2=('VGG_PONG_LAYERNECK', 1, 16, 4, 84, 256, True),
                                        c5=('F', 3, 64, 30, 256, 16, False))[cfg]
    O, plans, P, Bf, xs, xt, mk = _transporter_case(typ, cin, z, K, size, B, pm1)
    r64 = oracle_step(O, O.transporter_train_step, plans, P, Bf, torch.float64, xs, xt)
    r32 = oracle_step(O, O.transporter_train_step, plans, P, Bf, torch.float32, xs, xt)
    net, opt = hip_net(mk, P, Bf, 'fp32')
    loss, out3, G, _ = _run_transporter(net, opt, xs, xt)
    check_fp32_step(net, out3, loss, G, P, r64, r32, noise_names(plans))


def test_train_step_bf16_full_batch_pong():
    """BASELINE config 2 as named (Pong-grey 84x84 K=4, bf16, 256 pairs): judged DIRECTLY by SURVEY 8c's 16-bit
    tolerance -- x_t <= 2e-2 max|x_t|, keypoints atol 5e-3, loss rtol 1e-2 -- against the fp64 oracle.
    Gradients: per-tensor rel-L2 <= 0.35 (measured: median 0.08, worst 0.25 on the keypoint net's first layer, whose
    gradient passes through the whole bf16 chain; the same tensors are at 1e-3 in fp32).  BatchNorm running statistics
    to 2e-3 (bf16 resolution of the batch mean)."""
    O, plans, P, Bf, xs, xt, mk = _transporter_case('VGG_PONG_LAYERNECK', 1, 16, 4, 84, 256, True)
    l64, o64, G64, P64, B64 = oracle_step(O, O.transporter_train_step, plans, P, Bf, torch.float64, xs, xt)
    net, opt = hip_net(mk, P, Bf, 'bf16')
    loss, (x_t, k, m), G, _ = _run_transporter(net, opt, xs, xt)
    assert maxerr(x_t, o64[0]) <= 2e-2 * float(o64[0].abs().max())
    assert maxerr(k, o64[2]) <= 5e-3
    assert abs(float(loss) - float(l64)) <= 1e-2 * float(l64)
    noise = noise_names(plans)
    errs = {n: rel(G[n], g) for n, g in G64.items() if n not in noise}
    assert max(errs.values()) <= 0.35 and np.median(list(errs.values())) <= 0.15, sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    sd = net.state_dict()
    for n, b64 in B64.items():
        if not n.endswith('num_batches_tracked'):
            assert maxerr(sd[n], b64) <= 2e-3 * max(1.0, float(b64.abs().max())), n


@pytest.mark.parametrize('cfg', ['c4', 'c5'])
def test_train_step_bf16_full_batch_F(cfg):
    """Model F in bf16 at the BASELINE batch (config 4: 128 pairs at 128x128 K=10 -- the bench workload; config 5
    shape: 64 pairs at 256x256 K=30), one train step against the fp64 oracle.

    What holds directly: loss rtol 1e-2 (measured 3.5e-4), BatchNorm statistics to bf16 resolution, every op-level
    kernel within one bf16 ulp (tests/test_gpu_conv_mfma.py, test_gpu_bn_fullsize.py).  What does NOT hold for an
    8-bit mantissa, and is stated here rather than hidden behind an emulation: SURVEY 8c's x_t <= 2e-2 max|x_t| and
    keypoints 5e-3 were set for apex-O2 fp16 (11-bit mantissa).  At random init this 25-layer BatchNorm network
    amplifies a perturbation by ~1.3x per layer (measured on the oracle, layer by layer): bf16 storage rounding (2^-9)
    arrives at the encoder output as 4.4e-2 rel-L2, moves keypoints by up to 2.6e-2, and exp(-d/0.02) in the Gaussian
    map turns that into O(1) changes of single map pixels; fp16 storage (precision 'fp16', the tests further down)
    is 8x tighter at every stage and meets the stated tolerance.  The bounds asserted for bf16 are the fp16 contract's
    x 8 (oracle.TOL16['bf16'], a fixed rule: 3 fewer mantissa bits): keypoints <= 4e-2, encoder outputs <= 7.2e-2 rel-L2,
    x_t rel-L2 <= 0.36 at random init (measured at this size: 2.6e-2, 4.4e-2, 0.21)."""
    typ, cin, z, K, size, B, pm1 = dict(c4=('F', 3, 64, 10, 128, 128, False), c5=('F', 3, 64, 30, 256, 64, False))[cfg]
    O, plans, P, Bf, xs, xt, mk = _transporter_case(typ, cin, z, K, size, B, pm1)
    l64, o64, G64, P64, B64 = oracle_step(O, O.transporter_train_step, plans, P, Bf, torch.float64, xs, xt)
    net, opt = hip_net(mk, P, Bf, 'bf16')
    loss, (x_t, k, m), G, out = _run_transporter(net, opt, xs, xt)
    T = O.TOL16['bf16']
    assert abs(float(loss) - float(l64)) <= T['loss'] * float(l64)
    assert rel(out[1], o64[1]) <= T['features_rel_l2'] and rel(out[5], o64[5]) <= T['features_rel_l2']      # phi, heat-maps
    assert maxerr(k, o64[2]) <= T['keypoints']
    assert rel(x_t, o64[0]) <= T['xt_rel_l2_random_init']
    assert float(k.min()) >= 0 and float(k.max()) <= 1
    for n, g in G.items():
        assert torch.isfinite(g).all(), n
    sd = net.state_dict()
    for n, b64 in B64.items():
        if not n.endswith('num_batches_tracked'):
            assert maxerr(sd[n], b64) <= 4e-3 * max(1.0, float(b64.abs().max())), n
    for n, p0 in P.items():   # Adam step 1 moves every parameter by at most lr
        assert float((sd[n].double() - p0.cuda().double()).abs().max()) <= 1.01 * LR


@pytest.mark.parametrize('cfg', ['c4', 'c5'])
def test_train_step_fp16_full_batch_F(cfg):
    """Model F in fp16 (the reference's own 16-bit type, with dynamic loss scaling) at the BASELINE batch -- config 5 AS
    NAMED ("fp16 mixed precision", 64 pairs at 256x256 K=30) and config 4 (128 pairs) -- against the fp64 oracle,
    judged directly: keypoints atol 5e-3 and loss rtol 1e-2 hold; encoder outputs within 9e-3 rel-L2.
    x_t: SURVEY's 2e-2 max|x_t| does not hold at random init even for fp16 -- the decoder sees maps exp(-d/0.02) whose
    keypoints moved by 3e-3 -- measured 3.0e-2 rel-L2 / 5.2e-2 max|x_t|; asserted <= 4.5e-2 / 8e-2, and no worse than
    1.25x a torch emulation of fp16 storage of the REFERENCE itself (oracle quant=fp16_storage), i.e. this is what
    apex O2 would do to the reference on this input.  Gradients (un-scaled) per tensor: median rel-L2 <= 0.35."""
    typ, cin, z, K, size, B, pm1 = dict(c4=('F', 3, 64, 10, 128, 128, False), c5=('F', 3, 64, 30, 256, 64, False))[cfg]
    O, plans, P, Bf, xs, xt, mk = _transporter_case(typ, cin, z, K, size, B, pm1)
    l64, o64, G64, P64, B64 = oracle_step(O, O.transporter_train_step, plans, P, Bf, torch.float64, xs, xt)
    le, oe, Ge, _, _ = oracle_step(O, O.transporter_train_step, plans, P, Bf, torch.float32, xs, xt, quant=O.fp16_storage)
    net, opt = hip_net(mk, P, Bf, 'fp16')
    loss, (x_t, k, m), G, out = _run_transporter(net, opt, xs, xt)
    assert opt.scaler.steps_skipped == 0 and opt.scaler.adam_steps == 1
    T = O.TOL16['fp16']
    assert abs(float(loss) - float(l64)) <= T['loss'] * float(l64)
    assert maxerr(k, o64[2]) <= T['keypoints']
    assert rel(out[1], o64[1]) <= T['features_rel_l2'] and rel(out[5], o64[5]) <= T['features_rel_l2']
    assert rel(x_t, o64[0]) <= T['xt_rel_l2_random_init'] and maxerr(x_t, o64[0]) <= T['xt_worst_random_init'] * float(o64[0].abs().max())
    assert rel(x_t, o64[0]) <= 1.25 * rel(oe[0], o64[0]) + 1e-3, (rel(x_t, o64[0]), rel(oe[0], o64[0]))
    assert maxerr(k, o64[2]) <= 1.5 * maxerr(oe[2], o64[2]) + 5e-4
    noise = noise_names(plans)
    errs = [rel(G[n], g) for n, g in G64.items() if n not in noise]
    assert np.median(errs) <= 0.35, np.median(errs)
    sd = net.state_dict()
    for n, b64 in B64.items():
        if not n.endswith('num_batches_tracked'):
            assert maxerr(sd[n], b64) <= 1e-3 * max(1.0, float(b64.abs().max())), n
    for n, p0 in P.items():
        assert float((sd[n].double() - p0.cuda().double()).abs().max()) <= 1.01 * LR


def test_train_step_fp16_full_batch_pong():
    """Config 2's net in fp16 at 256 pairs: SURVEY 8c's 16-bit tolerance directly, with a wide margin (measured x_t 2.1e-3
    max|x_t|, keypoints 1.6e-5, loss 1.3e-5; gradients median 3e-2)."""
    O, plans, P, Bf, xs, xt, mk = _transporter_case('VGG_PONG_LAYERNECK', 1, 16, 4, 84, 256, True)
    l64, o64, G64, P64, B64 = oracle_step(O, O.transporter_train_step, plans, P, Bf, torch.float64, xs, xt)
    net, opt = hip_net(mk, P, Bf, 'fp16')
    loss, (x_t, k, m), G, _ = _run_transporter(net, opt, xs, xt)
    assert maxerr(x_t, o64[0]) <= 5e-3 * float(o64[0].abs().max())
    assert maxerr(k, o64[2]) <= 1e-4
    assert abs(float(loss) - float(l64)) <= 1e-3 * float(l64)
    noise = noise_names(plans)
    errs = [rel(G[n], g) for n, g in G64.items() if n not in noise]
    assert max(errs) <= 0.15 and np.median(errs) <= 0.06, (max(errs), np.median(errs))


def test_loss_scaler_skips_on_overflow_and_grows():
    """keypoints_amd.amp (the role of apex amp.scale_loss, transporter.py:84-86): a scale that overflows fp16 gradients
    makes the step a no-op (parameters and Adam state untouched, step count not advanced) and halves the scale; after
    `growth_interval` clean steps the scale doubles.  Everything is decided on the device."""
    import keypoints_amd as ka
    O, plans, P, Bf, xs, xt, mk = _transporter_case('VGG_PONG_LAYERNECK', 1, 16, 4, 84, 32, True)
    net, opt = hip_net(mk, P, Bf, 'fp16', init_scale=2.0 ** 40, growth_interval=2, max_scale=2.0 ** 60)
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    skipped = 0
    for it in range(40):
        opt.zero_grad()
        out = net(xs, xt)
        loss = ka.l2_reconstruction_loss(out[0], xt)
        with ka.amp.scale_loss(loss, opt) as scaled:
            scaled.backward()
        scale_was = opt.scaler.scale
        opt.step()
        if opt.scaler.steps_skipped > skipped:                     # overflow: nothing moved, scale halved
            skipped = opt.scaler.steps_skipped
            assert opt.scaler.scale == scale_was / 2 and opt.scaler.adam_steps == 0
            assert all(torch.equal(p, before[n]) for n, p in net.named_parameters())
        else:
            break
    assert 1 <= skipped < 39, skipped                              # 2^40 overflows; some power of two below does not
    assert opt.scaler.adam_steps == 1
    assert any(not torch.equal(p, before[n]) for n, p in net.named_parameters())
    s0 = opt.scaler.scale
    for _ in range(2):                                             # growth_interval = 2 clean steps (the first is done)
        opt.zero_grad()
        loss = ka.l2_reconstruction_loss(net(xs, xt)[0], xt)
        with ka.amp.scale_loss(loss, opt) as scaled:
            scaled.backward()
        opt.step()
    assert opt.scaler.scale in (s0, 2 * s0, s0 / 2)                # grew once (or backed off again right after growing)
    assert opt.scaler.adam_steps + opt.scaler.steps_skipped - skipped == 3


@pytest.mark.parametrize('precision,bound', [('bf16', 0.04), ('fp16', 0.03)])
def test_loss_trajectory_16bit_tracks_fp32_oracle(precision, bound):
    """20 train steps from the same init on the same stream of fresh batches (config 4, 128 pairs): the 16-bit HIP loss
    curve against the fp32 oracle's.  Per-step gradients differ (see above), the optimisation does not: every loss within
    4 % in bf16 (measured: <= 2.3 %) and 3 % in fp16 (measured: <= 2.1 %; late in the run both are dominated by the
    divergence of two trajectories, not by the rounding), the curve falls by more than 10x in both."""
    import keypoints_amd as ka
    O, plans, P, Bf, _, _, mk = _transporter_case('F', 3, 64, 10, 128, 128, False)
    net, opt = hip_net(mk, P, Bf, precision)
    oP = {k: v.cuda().clone() for k, v in P.items()}
    oB = {k: v.cuda().clone() for k, v in Bf.items()}
    st = {}
    g = torch.Generator().manual_seed(6)
    hip, ref = [], []
    for _ in range(20):
        a = torch.rand(128, 3, 128, 128, generator=g).cuda()
        b = torch.rand(128, 3, 128, 128, generator=g).cuda()
        opt.zero_grad()
        loss = ka.l2_reconstruction_loss(net(a, b)[0], b)
        with ka.amp.scale_loss(loss, opt) as scaled:
            scaled.backward()
        opt.step()
        with native_conv():
            ol, _, _ = O.transporter_train_step(plans, oP, oB, st, a, b)
        hip.append(float(loss)); ref.append(float(ol))
    hip, ref = np.array(hip), np.array(ref)
    assert np.all(np.abs(hip - ref) <= bound * ref), np.abs(hip - ref) / ref
    assert hip[-1] < 0.1 * hip[0] and ref[-1] < 0.1 * ref[0]


def test_keynet_tps_train_step_fp32_full_batch():
    """BASELINE config 3 (KeyNet F 128x128 K=10 + TPS/rotation augmentation, 128 images): the augmentation kernels
    against the oracle's grid_sample chain on the same drawn parameters (atol 4e-5: tests/test_tps.py's 1e-5 was set on
    32x32 images; one ulp of a normalised sampling coordinate is size/2 * 6e-8 pixels, so the interpolation error grows
    with the image size -- measured 2.1e-5 at 128x128), then
    one strict-fp32 train step on the augmented batch against the fp64 oracle."""
    import types
    import keypoints_amd as ka
    from keypoints_amd import data_augments as DA
    from keypoints_amd.models import keynet
    from oracle import kp_oracle as O
    B = 128
    plans = O.keynet_plans('F', 3, 64, 10)
    P, Bf = O.init_params(plans, seed=4)
    g = torch.Generator().manual_seed(31)
    x0 = torch.rand(B, 3, 128, 128, generator=g)
    # configs/keypoints_celeba.yaml:13-16: 4 control points, variance 0.05, max rotation 0.1 -- drawn as the reference
    # draws them (tps.py:122-125, data_augments.py:6-10)
    torch.manual_seed(77)
    p1 = DA.rand_peturb_params(B, 4, 0.05, 0.1)
    p2 = DA.rand_peturb_params(B, 4, 0.05, 0.1)
    ox, ox_, omask = O.tps_and_rotate(x0, p1, p2)
    xd = x0.cuda()
    mask = torch.ones_like(xd)
    x1 = DA.peturb(xd, *[t.cuda() for t in p1]); mask = DA.peturb(mask, *[t.cuda() for t in p1])
    x2 = DA.peturb(x1, *[t.cuda() for t in p2]); mask = DA.peturb(mask, *[t.cuda() for t in p2])
    assert maxerr(x1.cpu(), ox) <= 4e-5 and maxerr(x2.cpu(), ox_) <= 4e-5 and maxerr(mask.cpu(), omask) <= 4e-5

    def mk():
        return keynet.make(types.SimpleNamespace(model_type='F', model_in_channels=3, model_z_channels=64,
                                                 model_keypoints=10, load=None, transfer_load=None))
    r64 = oracle_step(O, O.keynet_train_step, plans, P, Bf, torch.float64, x1, x2, mask)
    r32 = oracle_step(O, O.keynet_train_step, plans, P, Bf, torch.float32, x1, x2, mask)
    net, opt = hip_net(mk, P, Bf, 'fp32')

    def f():
        out = net(x1, x2)
        return ka.l2_reconstruction_loss(out[0], x2, mask), out
    loss, out, G = hip_step(net, opt, f)
    check_fp32_step(net, (out[0], out[2], out[3]), loss, G, P, r64, r32, noise_names(plans))


def _keynet_case(B, seed=4):
    """BASELINE config 3's inputs: KeyNet F 128x128 K=10 on a TPS + rotation augmented batch (configs/keypoints_celeba.yaml:13-16),
    the augmentation run by the HIP warp kernels on parameters drawn as the reference draws them."""
    import types
    from keypoints_amd import data_augments as DA
    from keypoints_amd.models import keynet
    from oracle import kp_oracle as O
    plans = O.keynet_plans('F', 3, 64, 10)
    P, Bf = O.init_params(plans, seed=seed)
    g = torch.Generator().manual_seed(31)
    x0 = torch.rand(B, 3, 128, 128, generator=g).cuda()
    torch.manual_seed(77)
    p1 = DA.rand_peturb_params(B, 4, 0.05, 0.1)
    p2 = DA.rand_peturb_params(B, 4, 0.05, 0.1)
    mask = torch.ones_like(x0)
    x1 = DA.peturb(x0, *[t.cuda() for t in p1]); mask = DA.peturb(mask, *[t.cuda() for t in p1])
    x2 = DA.peturb(x1, *[t.cuda() for t in p2]); mask = DA.peturb(mask, *[t.cuda() for t in p2])

    def mk():
        return keynet.make(types.SimpleNamespace(model_type='F', model_in_channels=3, model_z_channels=64,
                                                 model_keypoints=10, load=None, transfer_load=None))
    return O, plans, P, Bf, x1, x2, mask, mk


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_keynet_tps_train_step_16bit_full_batch(precision):
    """BASELINE config 3 at its bench size (KeyNet F 128x128 K=10 + TPS / rotation, 128 images) in both 16-bit modes, one train
    step against the fp64 oracle on the same augmented batch.  Same reading as model F's Transporter above: loss rtol 1e-2
    and BatchNorm statistics hold directly in both modes; fp16 meets the contract's keypoint clause (5e-3) at random init; bf16 is
    held to the same clauses x 8 (oracle.TOL16).  The full contract, x_t clause included, is asserted on the conditioned network
    (test_conditioned_networks_meet_the_16bit_contract[c3])."""
    import keypoints_amd as ka
    O, plans, P, Bf, x1, x2, mask, mk = _keynet_case(128)
    l64, o64, G64, P64, B64 = oracle_step(O, O.keynet_train_step, plans, P, Bf, torch.float64, x1, x2, mask)
    net, opt = hip_net(mk, P, Bf, precision)

    def f():
        out = net(x1, x2)
        return ka.l2_reconstruction_loss(out[0], x2, mask), out
    loss, out, G = hip_step(net, opt, f)
    x_t, zz, k = out[0], out[1], out[2]
    m = dict(loss=abs(float(loss) - float(l64)) / float(l64), k=maxerr(k, o64[2]), z=rel(zz, o64[1]), x_t=rel(x_t, o64[0]),
             x_t_max=maxerr(x_t, o64[0]) / float(o64[0].abs().max()))
    print(f'config 3 {precision} 128 images:', {a: f'{b:.2e}' for a, b in m.items()})
    T = O.TOL16[precision]    # random-init network: the contract's keypoint / loss / feature clauses, x_t at its random-init number
    assert m['loss'] <= T['loss'] and m['k'] <= T['keypoints'] and m['z'] <= T['features_rel_l2'] and m['x_t'] <= T['xt_rel_l2_random_init'], m
    if opt.scaler is not None:
        assert opt.scaler.steps_skipped == 0 and opt.scaler.adam_steps == 1
    noise = noise_names(plans)
    errs = [rel(G[n], g) for n, g in G64.items() if n not in noise]
    assert all(torch.isfinite(g).all() for g in G.values()) and np.median(errs) <= (0.6 if precision == 'bf16' else 0.35), np.median(errs)
    sd = net.state_dict()
    for n, b64 in B64.items():
        if not n.endswith('num_batches_tracked'):
            assert maxerr(sd[n], b64) <= (4e-3 if precision == 'bf16' else 1e-3) * max(1.0, float(b64.abs().max())), n
    for n, p0 in P.items():
        assert float((sd[n].double() - p0.cuda().double()).abs().max()) <= 1.01 * LR


def test_train_step_fp16_config5_named_size():
    """BASELINE config 5 AS NAMED: model F, 256x256x3, K=30, fp16 mixed precision, 256 pairs per GPU (2048 over 8 GPUs) -- a
    112-136 GB working set on the HIP side.  An autograd oracle does not fit at this size (fp64 activations of one backward:
    > 1 TB), so:
      * forward, loss and BatchNorm running statistics against the fp64 oracle's FORWARD on the whole batch (no_grad: one
        layer's activations at a time; BatchNorm couples the batch, so it cannot be chunked) at the tolerances of the 64-pair
        case above;
      * the gradient by a size-independent property: the directional derivative of the fp64 oracle's loss along the HIP
        gradient, (L(p + e u) - L(p - e u)) / 2e with u = g / |g|, equals |g| when g is the gradient (two more oracle
        forwards); asserted within [0.8, 1.1] |g| (a direction 35 degrees off would read 0.82) -- printed for calibration;
      * Adam moves every parameter by at most lr, the loss scaler neither overflowed nor skipped."""
    import keypoints_amd as ka
    from keypoints_amd.models import transporter
    from oracle import kp_oracle as O
    B, size, K = 256, 256, 30
    plans = O.transporter_plans('F', 3, 64, K)
    P, Bf = O.init_params(plans, seed=3)
    g = torch.Generator().manual_seed(21)
    xs = torch.rand(B, 3, size, size, generator=g).cuda()
    xt = torch.rand(B, 3, size, size, generator=g).cuda()

    def oracle_forward(Pd):
        oB = {k: (v.to('cuda', torch.float64) if v.dtype.is_floating_point else v.to('cuda')).clone() for k, v in Bf.items()}
        with torch.no_grad(), native_conv():
            out = O.transporter_forward(plans, Pd, xs.double(), xt.double(), oB, True)
            loss = O.l2_reconstruction_loss(out[0], xt.double())
        return loss, out, oB

    P64 = {k: v.to('cuda', torch.float64) for k, v in P.items()}
    l64, o64, B64 = oracle_forward(P64)
    ref = dict(x_t=o64[0].float(), phi=o64[1].float(), k=o64[2].float(), heat=o64[5].float())
    x_max = float(o64[0].abs().max())
    del o64
    torch.cuda.empty_cache()

    net, opt = hip_net(lambda: transporter.make('F', 3, 64, K), P, Bf, 'fp16')
    loss, (x_t, k, m), G, out = _run_transporter(net, opt, xs, xt)
    assert opt.scaler.steps_skipped == 0 and opt.scaler.adam_steps == 1
    meas = dict(loss=abs(float(loss) - float(l64)) / float(l64), k=maxerr(k, ref['k']), phi=rel(out[1], ref['phi']),
                heat=rel(out[5], ref['heat']), x_t=rel(x_t, ref['x_t']), x_t_max=maxerr(x_t, ref['x_t']) / x_max)
    print('config 5, fp16, 256 pairs:', {a: f'{b:.2e}' for a, b in meas.items()})
    assert meas['loss'] <= 1e-2 and meas['k'] <= 5e-3 and meas['phi'] <= 9e-3 and meas['heat'] <= 9e-3, meas
    assert meas['x_t'] <= 4.5e-2 and meas['x_t_max'] <= 8e-2, meas
    sd = net.state_dict()
    for n, b64 in B64.items():
        if n.endswith('num_batches_tracked'):
            assert int(sd[n]) == int(b64)
        else:
            assert maxerr(sd[n], b64) <= 1e-3 * max(1.0, float(b64.abs().max())), n
    for n, p0 in P.items():
        assert float((sd[n].double() - p0.cuda().double()).abs().max()) <= 1.01 * LR
        assert torch.isfinite(G[n]).all(), n
    del net, opt, out, x_t, m, ref
    torch.cuda.empty_cache()
    # directional derivative of the fp64 oracle's loss along the HIP gradient
    noise = noise_names(plans)
    gn = float(torch.sqrt(sum((G[n].double() ** 2).sum() for n in G if n not in noise)))
    eps = 1e-3 * float(l64) / gn     # a step that changes the loss by ~0.1 % each way (central difference: even orders cancel)
    Lp, _, _ = oracle_forward({n: (p + eps * G[n].double() / gn if n not in noise else p) for n, p in P64.items()})
    Lm, _, _ = oracle_forward({n: (p - eps * G[n].double() / gn if n not in noise else p) for n, p in P64.items()})
    d = float(Lp - Lm) / (2 * eps)
    print(f'config 5 directional derivative / |g| = {d / gn:.4f}  (|g| = {gn:.4e}, eps = {eps:.3e})')
    assert 0.8 <= d / gn <= 1.1, (d, gn)


def _condition(make_net, step_fn, steps, check=0.05):
    """`steps` Adam steps of the strict-fp32 HIP path on fresh structured batches; -> the conditioned state dict (CPU)."""
    import keypoints_amd as ka
    net = make_net().cuda().set_precision('fp32')
    opt = ka.FusedAdam(net.parameters(), lr=LR).bind(net)
    first = last = None
    for it in range(steps):
        opt.zero_grad()
        loss = step_fn(net, it)
        loss.backward()
        opt.step()
        if it == 0:
            first = float(loss)
        if os.environ.get('KP_COND_TRACE') and it in (1, 2, 5, 10, 20, 50, 100, 150):
            print(f'  it {it}: {float(loss):.6e}')
    last = float(loss)
    print(f'conditioning: loss {first:.6e} -> {last:.6e}')
    assert last < check * first, (first, last)            # the network did get conditioned
    return {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}


@pytest.mark.parametrize('cfg', ['c4', 'c5', 'c3'])
def test_conditioned_networks_meet_the_16bit_contract(cfg):
    """THE 16-bit contract (oracle.TOL16), asserted exactly as written there, for the networks of BASELINE
    configs 4 (Transporter F 128x128 K=10, 128 pairs), 5 (F 256x256 K=30; 32 pairs: the fp64 oracle step of a larger batch does not
    fit next to the HIP step) and 3 (KeyNet F 128x128 K=10 + TPS / rotation, 128 images): the network is conditioned with the
    strict-fp32 HIP path on structured pairs (tests/structured.py), the SAME weights then go to the fp64 oracle and to fresh fp16 /
    bf16 HIP networks, one train-mode step each.  fp16 (+ dynamic loss scaling: the reference's O2 recipe, transporter.py:51,84-86,
    configs/transporter_celeba.yaml:7, configs/keypoints_celeba.yaml:7) is the contract type and bench.py's headline dtype; bf16 is
    held to the same clauses x 8 (3 fewer mantissa bits; keypoints x 16: oracle.TOL16).

    Measured, config 4 (tools/probes/conditioned.py, 0 / 50 / 200 / 600 steps): conditioning shrinks the rel-L2 distances 10-20x
    (x_t 2.1e-1 -> 1.4e-2 bf16, 3.3e-2 -> 1.8e-3 fp16) but NOT the keypoint error, which is storage rounding through ten BatchNorm
    layers whatever the weights: bf16 1.9e-2 - 3.6e-2, fp16 3.4e-3 - 3.9e-3."""
    import types
    import keypoints_amd as ka
    from keypoints_amd import data_augments as DA
    from keypoints_amd.models import keynet, transporter
    from oracle import kp_oracle as O
    from structured import structured_pairs
    import os
    # Three conditioning trajectories (seeds of the structured batches; KP_COND_SEED: one, for the probes tools/probes/contract_*).
    # The contract is a statement about the 16-bit arithmetic, and a single conditioned network scatters around it: over six
    # trajectories the fp16 keypoint error of config 3 is 3.95 / 4.09 / 4.14 / 4.21 / 5.07 / 6.00 e-3 (stated: 5e-3), of config 4
    # 2.74 / 2.76 / 3.26 / 5.13 e-3 (profiles/r04_contract_spread.txt, r04i_contract_spread_c3.txt) -- which network a trajectory
    # ends in depends on the last bits of the strict-fp32 path (any re-ordering of an fp32 sum moves it).  Asserted: every clause on
    # the MEDIAN of the three networks, and every single network inside oracle.TOL16_PER_NETWORK (1.5) x the clause.
    seeds = [int(os.environ['KP_COND_SEED'])] if os.environ.get('KP_COND_SEED') else [99, 100, 101]
    if cfg == 'c3':
        plans = O.keynet_plans('F', 3, 64, 10)
        mk = lambda: keynet.make(types.SimpleNamespace(model_type='F', model_in_channels=3, model_z_channels=64, model_keypoints=10,
                                                       load=None, transfer_load=None))
        size, tb, nev, steps = 128, 32, 128, 200
    else:
        K, size, tb, nev, steps = dict(c4=(10, 128, 32, 128, 200), c5=(30, 256, 8, 32, 160))[cfg]
        plans = O.transporter_plans('F', 3, 64, K)
        mk = lambda: transporter.make('F', 3, 64, K)
    P, Bf = O.init_params(plans, seed=3)

    def make_net():
        n = mk()
        n.load_state_dict({**P, **Bf}, strict=True)
        return n

    xs0, xt0 = structured_pairs(nev, size, torch.Generator().manual_seed(7), device='cuda')
    results = {'fp16': [], 'bf16': []}
    emus = {'fp16': [], 'bf16': []}
    for seed in seeds:
        g = torch.Generator().manual_seed(seed)

        def train(net, it):
            a, b = structured_pairs(tb, size, g, device='cuda')
            return ka.l2_reconstruction_loss(net(a, b)[0], b)
        if os.environ.get('KP_COND_LOAD'):
            sd = torch.load(os.environ['KP_COND_LOAD'])
        else:
            sd = _condition(make_net, train, steps)
        if os.environ.get('KP_COND_SAVE'):
            torch.save(sd, os.environ['KP_COND_SAVE'])
        Pn, Bn = {k: sd[k] for k in P}, {k: sd[k] for k in Bf}
        xs, xt = xs0, xt0
        mask = None
        if cfg == 'c3':       # config 3's input pipeline: two random TPS + rotation warps of the batch (keypoints/data_augments.py:27-38)
            torch.manual_seed(77)
            p1 = DA.rand_peturb_params(nev, 4, 0.05, 0.1)
            p2 = DA.rand_peturb_params(nev, 4, 0.05, 0.1)
            mask = torch.ones_like(xs)
            x1 = DA.peturb(xs, *[t.cuda() for t in p1]); mask = DA.peturb(mask, *[t.cuda() for t in p1])
            x2 = DA.peturb(x1, *[t.cuda() for t in p2]); mask = DA.peturb(mask, *[t.cuda() for t in p2])
            xs, xt = x1, x2
            l64, o64, G64, P64, B64 = oracle_step(O, O.keynet_train_step, plans, Pn, Bn, torch.float64, xs, xt, mask)
        else:
            l64, o64, G64, P64, B64 = oracle_step(O, O.transporter_train_step, plans, Pn, Bn, torch.float64, xs, xt)
        xmax = float(o64[0].abs().max())

        def clauses(loss, x_t, feat, k, T):
            err = (x_t.double() - o64[0]).abs() / xmax
            return dict(loss=abs(float(loss) - float(l64)) / float(l64), keypoints=maxerr(k, o64[2]), features_rel_l2=rel(feat, o64[1]),
                        xt_rel_l2=rel(x_t, o64[0]), xt_worst=float(err.max()), xt_frac_beyond=float((err > T['xt_beyond']).double().mean()))
        for precision in ('fp16', 'bf16'):
            T = O.TOL16[precision]
            # the REFERENCE's own arithmetic with this storage type emulated (oracle quant=*_storage: fp32 sums, operands and stored
            # activations rounded where apex O2 / the HIP engine store them), same conditioned weights, same batch, against the same
            # fp64 run: what the storage type costs the reference itself on THIS network
            qf = O.fp16_storage if precision == 'fp16' else O.bf16_storage
            if cfg == 'c3':
                le, oe, _, _, _ = oracle_step(O, O.keynet_train_step, plans, Pn, Bn, torch.float32, xs, xt, mask, quant=qf)
            else:
                le, oe, _, _, _ = oracle_step(O, O.transporter_train_step, plans, Pn, Bn, torch.float32, xs, xt, quant=qf)
            emu = clauses(le, oe[0], oe[1], oe[2], T)
            del le, oe
            print(f'emulation   {cfg}, {precision}, {nev} pairs, seed {seed}:', {a_: f'{b_:.2e}' for a_, b_ in emu.items()})
            net, opt = hip_net(mk, Pn, Bn, precision)

            def f():
                out = net(xs, xt)
                return ka.l2_reconstruction_loss(out[0], xt, mask), out
            loss, out, G = hip_step(net, opt, f)
            meas = clauses(loss, out[0], out[1], out[2], T)
            print(f'conditioned {cfg}, {precision}, {nev} pairs, seed {seed}:', {a_: f'{b_:.2e}' for a_, b_ in meas.items()})
            print(f'hip / emu   {cfg}, {precision}, {nev} pairs, seed {seed}:', {a_: f'{meas[a_] / max(emu[a_], 1e-30):.2f}' for a_ in meas})
            results[precision].append(meas)
            emus[precision].append(emu)
            if opt.scaler is not None:
                assert opt.scaler.steps_skipped == 0
            del net, opt
        del l64, o64, G64, P64, B64
    for precision, rows in results.items():
        T = O.TOL16[precision]
        for key in rows[0]:
            vals = sorted(r[key] for r in rows)
            med = vals[len(vals) // 2]
            assert med <= T[key], (cfg, precision, key, 'median', vals, T[key])
            assert vals[-1] <= min(1.0, O.TOL16_PER_NETWORK * T[key]) or T[key] >= 1.0, (cfg, precision, key, 'worst network', vals, T[key])
        # ... and, per clause and per NETWORK, against the reference's own arithmetic in this storage type on the same weights (the
        # emulation leg above): the HIP path may be no further from fp64 than TOL16_VS_EMULATION x the emulated reference is, plus
        # TOL16_EMULATION_FLOOR x the clause for statistics that are ~0 in both (loss error 1e-5, a handful of pixels beyond 2e-2).
        # This is the anchor of the contract: the absolute clauses say what 16-bit storage costs on this network family, this
        # one says the HIP path adds nothing to what the storage type costs the reference itself.
        # Measured (profiles/r05a_contract_emulation.txt, 3 configs x 3 networks x 2 types): keypoints 0.89 - 1.12, features /
        # x_t rel-L2 0.97 - 1.02, worst pixel 0.87 - 1.25 of the emulation.
        for r, e in zip(rows, emus[precision]):
            for key in r:
                assert r[key] <= O.TOL16_VS_EMULATION * e[key] + O.TOL16_EMULATION_FLOOR * T[key], (cfg, precision, key, 'vs emulation', r[key], e[key])
    # ---- the PREMISE of the re-scoped x_t clause and SURVEY's keypoint clause without any factor (round 6), fp16 =
    # the contract type.  (a) SURVEY 8c wrote "x_t <= 2e-2 max|x_t|"; oracle.TOL16 replaces it by a distribution statement because the
    # REFERENCE'S OWN arithmetic in fp16 storage breaks it on these networks.  That is asserted here, not assumed: wherever the HIP path
    # breaks the as-written clause, the emulated reference is at (>= SURVEY_XT_NEAR x) the threshold itself, and the median network of
    # the emulation breaks it whenever the median network of the HIP path does.  (b) SURVEY's keypoint clause 5e-3 with no per-network
    # factor holds on at least two of three networks.
    rows, es = results['fp16'], emus['fp16']
    hw, ew = [r['xt_worst'] for r in rows], [e['xt_worst'] for e in es]
    hip_v, emu_v = [v > O.SURVEY_XT_MAX for v in hw], [v > O.SURVEY_XT_MAX for v in ew]
    inside = [r['keypoints'] <= O.TOL16['fp16']['keypoints'] for r in rows]
    print(f'premise     {cfg}, fp16: as-written x_t clause (2e-2 max) broken by HIP {hip_v} / by the emulated reference {emu_v}; worst pixel '
          f'hip {[f"{v:.3f}" for v in hw]} emu {[f"{v:.3f}" for v in ew]}; keypoints inside 5e-3 (no factor): {sum(inside)} of {len(inside)}')
    for h, e, hv in zip(hw, ew, hip_v):
        assert (not hv) or e >= O.SURVEY_XT_NEAR * O.SURVEY_XT_MAX, (cfg, 'the HIP path breaks SURVEY\'s as-written x_t clause where the emulated reference is not near it', h, e)
    if len(rows) >= 3:
        med = lambda v: sorted(v)[len(v) // 2]
        assert med(hw) <= O.SURVEY_XT_MAX or med(ew) > O.SURVEY_XT_MAX, (cfg, 'median network: HIP breaks the as-written x_t clause, the emulated reference does not', hw, ew)
        assert sum(inside) >= O.SURVEY_KEYPOINT_QUORUM * len(inside) - 1e-9, (cfg, 'keypoints: fewer than two of three networks inside SURVEY\'s 5e-3', [r['keypoints'] for r in rows])
