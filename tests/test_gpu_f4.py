"""the non-default combine modes on their HIP blend kernels (forward + backward) against the
reference's vectors, the MaskMaker-wired 'pretrained_network' model, and the autoencoder pre-training checkpoint
path (autoencode.py:59-69 -> transporter.py:72-76 `load_from_autoencoder`)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def close(a, b, atol=1e-6, rtol=1e-5):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b)
    assert np.all(err <= atol + rtol * np.abs(b)), f'max err {err.max():.3e}'


@pytest.mark.parametrize('mode', ['max', 'sum_and_clamp', 'loop', 'pretrained_network'])
def test_transport_modes_forward_backward_golden(golden, mode):
    """functional.transport(mode): phi, both masks and all four input gradients against the reference's forward text
    replayed by tools/make_golden.py (transporter.py:41-60)."""
    from keypoints_amd.models import functional as MF
    z = golden('transport.npz')
    a = [torch.tensor(z[n], device='cuda', requires_grad=True) for n in ('phi_s', 'phi_t', 'm_s', 'm_t')]
    ms, mt = (a[2], a[3]) if mode != 'pretrained_network' else (a[2][:, :1], a[3][:, :1])
    phi, mask_s, mask_t = MF.transport(a[0], a[1], ms, mt, mode)
    close(phi, z[f'{mode}.phi']); close(mask_s, z[f'{mode}.mask_s']); close(mask_t, z[f'{mode}.mask_t'])
    (phi * torch.tensor(z['w'], device='cuda')).sum().backward()
    for t, n in zip(a, ('d_phi_s', 'd_phi_t', 'd_m_s', 'd_m_t')):
        close(t.grad, z[f'{mode}.{n}'], atol=1e-5)


@pytest.mark.parametrize('mode', ['sum_and_clamp', 'loop', 'pretrained_network'])
def test_transport_modes_vs_oracle_autograd_bigger(mode):
    """Config-4-sized tensors (16x16, z=64, K=10) incl. sums that leave [0, 1] (clamp's dead zone) vs the oracle in fp64."""
    from keypoints_amd.models import functional as MF
    from oracle import kp_oracle as O
    g = torch.Generator().manual_seed(4)
    B, C, K, H, W = 6, 64, 10, 16, 16
    phi_s, phi_t = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    Km = 1 if mode == 'pretrained_network' else K
    m_s = O.gaussian_like(torch.rand(B, Km, 2, generator=g), H, W) * 1.5      # overlapping blobs: sum > 1 in places
    m_t = O.gaussian_like(torch.rand(B, Km, 2, generator=g), H, W) * (1.5 if mode == 'sum_and_clamp' else 1.0)
    w = torch.randn(B, C, H, W, generator=g)
    d = [t.cuda().requires_grad_(True) for t in (phi_s, phi_t, m_s, m_t)]
    phi, ms, mt = MF.transport(*d, mode)
    (phi * w.cuda()).sum().backward()
    o = [t.double().requires_grad_(True) for t in (phi_s, phi_t, m_s, m_t)]
    ophi, oms, omt = O.transport(*o, mode)
    (ophi * w.double()).sum().backward()
    close(phi, ophi, atol=1e-5); close(ms, oms); close(mt, omt)
    for got, ref in zip(d, o):
        close(got.grad, ref.grad, atol=2e-5 * float(ref.grad.abs().max()), rtol=1e-4)


@pytest.mark.parametrize('K', [1, 3, 4, 5, 16, 17, 40])
def test_loop_mode_every_register_bound(K):
    """'loop' (the reference's default combine_method, transporter.py:14,41-45) runs on three instantiations of the blend kernels --
    K <= 4 and K <= 16 with the per-keypoint values in registers, up to 64 otherwise: each against the oracle's autograd in fp64,
    at both sides of the two bounds."""
    from keypoints_amd.models import functional as MF
    from oracle import kp_oracle as O
    g = torch.Generator().manual_seed(40 + K)
    B, C, H, W = 3, 16, 12, 20
    phi_s, phi_t = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    m_s = O.gaussian_like(torch.rand(B, K, 2, generator=g), H, W)
    m_t = O.gaussian_like(torch.rand(B, K, 2, generator=g), H, W)
    w = torch.randn(B, C, H, W, generator=g)
    d = [t.cuda().requires_grad_(True) for t in (phi_s, phi_t, m_s, m_t)]
    phi, ms, mt = MF.transport(*d, 'loop')
    (phi * w.cuda()).sum().backward()
    o = [t.double().requires_grad_(True) for t in (phi_s, phi_t, m_s, m_t)]
    ophi, oms, omt = O.transport(*o, 'loop')
    (ophi * w.double()).sum().backward()
    close(phi, ophi, atol=1e-5); close(ms, oms); close(mt, omt)
    for got, ref in zip(d, o):
        close(got.grad, ref.grad, atol=2e-5 * float(ref.grad.abs().max()) + 1e-7, rtol=1e-4)


def test_pretrained_network_model_wires_maskmaker():
    """transporter.make(combine_mode='pretrained_network'): key2map = MaskMaker(mapper) -- keypoints -> point_map ->
    frozen mapper Unit (mapper.py:50-52) -> ONE mask channel; forward equals composing the pieces by hand with the
    oracle's blend, and a train step moves the encoder / decoder but not the mapper."""
    import keypoints_amd as ka
    from keypoints_amd.models import functional as MF, transporter
    from oracle import kp_oracle as O
    torch.manual_seed(3)
    net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 3, combine_mode='pretrained_network').cuda()
    assert isinstance(net.key2map, transporter.MaskMaker)
    mapper_before = [p.detach().clone() for p in net.key2map.map.parameters()]
    assert mapper_before and not any(p.requires_grad for p in net.key2map.map.parameters())
    opt = ka.FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4).bind(net)
    g = torch.Generator().manual_seed(5)
    xs = (torch.rand(4, 1, 32, 32, generator=g) * 2 - 1).cuda()
    xt = (torch.rand(4, 1, 32, 32, generator=g) * 2 - 1).cuda()
    net.eval()
    with torch.no_grad():
        out = net(xs, xt)
        phi_s, _, k_s, _, m_s = net.extract(xs)
        phi_t, _, k_t, _, m_t = net.extract(xt)
        assert m_s.shape == (4, 1, 32, 32)
        # (MaskMaker runs the frozen eval-mode mapper as ONE kernel with BatchNorm folded into its weights -- fp32, another rounding
        #  order than the layer chain it is compared with here)
        chain = net.key2map.map(MF.point_map(k_t, 32, 32))
        assert net.key2map._fused_unit is not None
        close(m_t, chain, atol=2e-5 * float(chain.abs().max()), rtol=1e-4)
        ophi, _, _ = O.transport(phi_s.cpu(), phi_t.cpu(), m_s.cpu(), m_t.cpu(), 'pretrained_network')
    close(out[1], ophi, atol=1e-5)
    net.train()
    w0 = net.decoder.in_block[1].weight.detach().clone()
    opt.zero_grad()
    loss = ka.l2_reconstruction_loss(net(xs, xt)[0], xt)
    loss.backward()
    opt.step()
    assert not torch.equal(net.decoder.in_block[1].weight, w0)
    assert all(torch.equal(p, q) for p, q in zip(net.key2map.map.parameters(), mapper_before))


def _write_ckpt(z, root):
    """`<root>/{encoder,decoder}/{in_block,core,out_block}.mdl` as the reference's Unit.save wrote them (knn.py:143-154)."""
    from collections import OrderedDict
    for unit in ('encoder', 'decoder'):
        os.makedirs(os.path.join(root, unit), exist_ok=True)
        for blk in ('in_block', 'core', 'out_block'):
            sd = OrderedDict((str(k), torch.tensor(z[f'ckpt.{unit}.{blk}.{k}'])) for k in z[f'files.{unit}.{blk}'])
            torch.save(sd, os.path.join(root, unit, blk + '.mdl'))


def test_autoencoder_loads_reference_checkpoint_and_transfers(golden, tmp_path):
    """A checkpoint written by the reference's Units (tests/golden/autoencoder_pong.npz holds the state dicts of the
    .mdl files) loads into keypoints_amd.models.autoencoder.AutoEncoder and reproduces the reference's z and
    reconstruction; TransporterNet.load_from_autoencoder takes exactly the blocks the reference takes
    (transporter.py:72-76); save() writes the same files back."""
    from keypoints_amd.models import autoencoder, transporter
    z = golden('autoencoder_pong.npz')
    root = str(tmp_path / 'ae')
    _write_ckpt(z, root)
    ae = autoencoder.make('VGG_PONG_LAYERNECK', 1, 16, load=root).cuda().eval()
    with torch.no_grad():
        zz, xr = ae(torch.tensor(z['x']).cuda())
    close(zz, z['z'], atol=2e-5, rtol=1e-4); close(xr, z['x_rec'], atol=2e-5, rtol=1e-4)
    out = str(tmp_path / 'ae_out')
    ae.save(out)
    for unit in ('encoder', 'decoder'):
        for blk in ('in_block', 'core', 'out_block'):
            sd = torch.load(os.path.join(out, unit, blk + '.mdl'))
            assert list(sd.keys()) == [str(k) for k in z[f'files.{unit}.{blk}']]
            for k, v in sd.items():
                assert np.array_equal(v.cpu().numpy(), z[f'ckpt.{unit}.{blk}.{k}']), (unit, blk, k)
    torch.manual_seed(8)
    net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 3, transfer_load=root)
    sd = net.state_dict()
    taken = [n for n in sd if (n.startswith('feature.in_block') or n.startswith('feature.core')
                               or n.startswith('keypoint.in_block') or n.startswith('keypoint.core')
                               or n.startswith('decoder.core') or n.startswith('decoder.out_block'))]
    assert taken
    for n in sd:   # the transferred blocks AND the re-initialised rest (same seed, same construction order) match the reference
        ref = z['transfer.' + n]
        if n in taken or sd[n].dtype != torch.float32:
            assert np.array_equal(sd[n].numpy(), ref), n
        else:
            close(sd[n], ref, atol=1e-7)


def test_checkpoint_files_hold_only_their_block_after_bind(tmp_path):
    """FusedAdam.bind re-homes every parameter as a view of one per-device buffer; Unit.save (knn.py:143-154's files) must still write
    each block's own tensors, not the whole buffer behind the view: file size ~ the block's parameter bytes, values round-trip."""
    import keypoints_amd as ka
    from keypoints_amd.models import transporter
    torch.manual_seed(3)
    net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 3).cuda()
    ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
    total = 4 * sum(p.numel() for p in net.parameters())
    out = str(tmp_path / 'ck')
    net.save(out)
    for unit, mod in (('encoder', net.feature), ('keypoint', net.keypoint), ('decoder', net.decoder)):
        for blk in ('in_block', 'core', 'out_block'):
            path = os.path.join(out, unit, blk + '.mdl')
            own = sum(v.numel() * v.element_size() for v in getattr(mod, blk).state_dict().values())
            size = os.path.getsize(path)
            assert size <= own + 16384, (unit, blk, size, own)       # (pickle framing: a few KB)
            assert size < total, (unit, blk, size, total)
            sd = torch.load(path)
            for k, v in getattr(mod, blk).state_dict().items():
                assert torch.equal(sd[k].cpu(), v.cpu()), (unit, blk, k)


def test_amp_initialize_mirrors_the_reference_recipe():
    """`amp.initialize(net, optim, opt_level=...)` + `with amp.scale_loss(loss, optim)` (transporter.py:51,84-86) on the
    native implementation: O0 -> strict fp32 Units and no scaler (scale_loss is the identity), O2 -> fp16 Units with a
    dynamic scaler, loss_scale=<number> -> a static one; three steps of each run and lower the loss."""
    import keypoints_amd as ka
    from keypoints_amd import amp
    from keypoints_amd.models import knn, transporter
    g = torch.Generator().manual_seed(5)
    xs = (torch.rand(8, 1, 32, 32, generator=g) * 2 - 1).cuda()
    xt = (torch.rand(8, 1, 32, 32, generator=g) * 2 - 1).cuda()
    for opt_level, loss_scale, want in (('O0', 'dynamic', 'fp32'), ('O2', 'dynamic', 'fp16'), ('O1', 128.0, 'fp16')):
        torch.manual_seed(2)
        net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 3).cuda()
        opt = ka.FusedAdam(net.parameters(), lr=1e-3).bind(net)
        net, opt = amp.initialize(net, opt, opt_level=opt_level, loss_scale=loss_scale)
        assert all(m.precision == want for m in net.modules() if isinstance(m, knn.Unit))
        assert (opt.scaler is None) == (want == 'fp32')
        losses = []
        for _ in range(3):
            opt.zero_grad()
            loss = ka.l2_reconstruction_loss(net(xs, xt)[0], xt)
            with amp.scale_loss(loss, opt) as scaled_loss:
                if opt.scaler is None:
                    assert scaled_loss is loss
                else:
                    assert abs(float(scaled_loss) - float(loss) * opt.scaler.scale) <= 1e-3 * abs(float(scaled_loss))
                scaled_loss.backward()
            opt.step()
            losses.append(float(loss))
        assert losses[-1] < losses[0], losses
        if opt.scaler is not None:
            assert opt.scaler.steps_skipped == 0 and opt.scaler.adam_steps == 3
            if loss_scale != 'dynamic':
                assert opt.scaler.scale == 128.0
