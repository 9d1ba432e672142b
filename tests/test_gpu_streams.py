"""Stream-level concurrency of the train step: weight gradients on a side stream of the Unit's backward,
the keypoint encoder on a side stream of the feature encoder's.  It changes WHEN kernels run, never what they compute."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(net_kind, streams, steps=3, accumulate=False):
    import keypoints_amd as ka
    from keypoints_amd.models import transporter, keynet
    ka.set_multi_stream(streams)
    try:
        g = torch.Generator().manual_seed(5)
        # (128 x 128: every layer on the deterministic kernels, as in test_train_step_is_bitwise_reproducible)
        xs = torch.rand(8, 3, 128, 128, generator=g).cuda()
        xt = torch.rand(8, 3, 128, 128, generator=g).cuda()
        torch.manual_seed(11)
        if net_kind == 'keynet':
            class A: model_type, model_in_channels, model_z_channels, model_keypoints = 'F', 3, 32, 5
            net = keynet.make(A()).cuda().set_precision('bf16')
        else:
            net = transporter.make('F', 3, 32, 5).cuda().set_precision('bf16')
        opt = ka.FusedAdam(net.parameters(), lr=1e-3).bind(net)
        losses = []
        for _ in range(steps):
            opt.zero_grad()
            loss = ka.l2_reconstruction_loss(net(xs, xt)[0], xt)
            loss.backward()
            if accumulate:   # a second backward into the same arenas before the optimiser step
                ka.l2_reconstruction_loss(net(xt, xs)[0], xs).backward()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        grads = torch.cat([p.grad.flatten() for p in net.parameters()]).clone()
        return torch.stack(losses), {k: v.clone() for k, v in net.state_dict().items()}, grads
    finally:
        ka.set_multi_stream(True)


@pytest.mark.parametrize('net_kind', ['transporter', 'keynet'])
@pytest.mark.parametrize('accumulate', [False, True])
def test_side_streams_change_nothing(net_kind, accumulate):
    """Three train steps with the side streams on and off: losses, gradients of the last step, parameters and BatchNorm
    buffers are bit-identical (also with two backwards accumulating into the arenas per step)."""
    l0, s0, g0 = _run(net_kind, False, accumulate=accumulate)
    l1, s1, g1 = _run(net_kind, True, accumulate=accumulate)
    assert torch.equal(l0, l1)
    assert torch.equal(g0, g1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_gradients_are_final_when_backward_returns():
    """`p.grad` read on the caller's stream right after backward() (no optimiser, no synchronize in between) sees the
    finished arenas: the autograd pass joins the side streams when it ends."""
    import keypoints_amd as ka
    from keypoints_amd.models import transporter
    g = torch.Generator().manual_seed(5)
    xs = torch.rand(16, 3, 128, 128, generator=g).cuda()
    xt = torch.rand(16, 3, 128, 128, generator=g).cuda()
    outs = []
    for streams in (False, True, True):
        ka.set_multi_stream(streams)
        torch.manual_seed(11)
        net = transporter.make('F', 3, 32, 5).cuda().set_precision('bf16')
        ka.l2_reconstruction_loss(net(xs, xt)[0], xt).backward()
        # consumer kernels enqueued immediately on the current stream
        outs.append(torch.stack([p.grad.double().abs().sum() for p in net.parameters()]))
    ka.set_multi_stream(True)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize('model', ['F', 'VGG_PONG_LAYERNECK'])
def test_graphed_train_step_equals_eager(model):
    """keypoints_amd.graph.GraphedTrainStep: the whole step (zero_grad, forward, loss, backward with its side streams as
    graph branches, Adam with device-resident step count) captured once and replayed on fresh batches gives the same
    parameters, bit for bit, as the eager loop on the same batches."""
    import keypoints_amd as ka
    from keypoints_amd.models import transporter
    cin, z, K, size, B = (3, 64, 10, 128, 8) if model == 'F' else (1, 16, 4, 84, 32)
    g = torch.Generator().manual_seed(7)
    batches = [(torch.rand(B, cin, size, size, generator=g).cuda(), torch.rand(B, cin, size, size, generator=g).cuda())
               for _ in range(6)]

    def build():
        torch.manual_seed(11)
        net = transporter.make(model, cin, z, K).cuda().set_precision('bf16')
        opt = ka.FusedAdam(net.parameters(), lr=1e-3).bind(net)
        return net, opt

    net_g, opt_g = build()
    step = ka.graph.GraphedTrainStep(net_g, opt_g, None, batches[0], warmup=2)   # 2 warm-up steps train on batch 0 (the capture itself runs nothing)
    losses_g = [step(xs, xt).clone() for xs, xt in batches[1:]]

    net_e, opt_e = build()
    opt_e.scaler = ka.amp.LossScaler('cuda', init_scale=1.0, growth_factor=1.0, backoff_factor=1.0, growth_interval=1 << 30,
                                     max_scale=1.0)
    opt_e.scaler.unit = True
    losses_e = []
    for i, (xs, xt) in enumerate([batches[0]] * 2 + batches[1:]):
        opt_e.zero_grad()
        loss = ka.l2_reconstruction_loss(net_e(xs, xt)[0], xt)
        loss.backward()
        opt_e.step()
        if i >= 2:
            losses_e.append(loss.detach().clone())
    torch.cuda.synchronize()
    assert torch.equal(torch.stack(losses_g), torch.stack(losses_e))
    for (n, a), (_, b) in zip(net_g.state_dict().items(), net_e.state_dict().items()):
        assert torch.equal(a, b), n
    assert opt_g.scaler.adam_steps == 7 and opt_g.scaler.steps_skipped == 0


def test_graphed_train_step_fp16_dynamic_loss_scale():
    """The fp16 recipe (dynamic loss scaling, overflow check, skipped steps) lives on the device, so it replays inside the
    graph: same parameters and scaler state as the eager loop, including a step that overflows and is skipped."""
    import keypoints_amd as ka
    from keypoints_amd.models import transporter
    g = torch.Generator().manual_seed(7)
    batches = [(torch.rand(8, 3, 128, 128, generator=g).cuda(), torch.rand(8, 3, 128, 128, generator=g).cuda()) for _ in range(5)]
    batches[3] = (batches[3][0] * 3e4, batches[3][1] * 3e4)   # a batch that overflows fp16 activations: the step must be skipped

    def build():
        torch.manual_seed(11)
        net = transporter.make('F', 3, 64, 10).cuda()
        opt = ka.FusedAdam(net.parameters(), lr=1e-3)
        net, opt = ka.amp.initialize(net, opt, opt_level='O2')
        return net, opt.bind(net)

    net_g, opt_g = build()
    step = ka.graph.GraphedTrainStep(net_g, opt_g, None, batches[0], warmup=2)
    for xs, xt in batches[1:]:
        step(xs, xt)
    net_e, opt_e = build()
    for xs, xt in [batches[0]] * 2 + batches[1:]:
        opt_e.zero_grad()
        loss = ka.l2_reconstruction_loss(net_e(xs, xt)[0], xt)
        with ka.amp.scale_loss(loss, opt_e) as scaled:
            scaled.backward()
        opt_e.step()
    torch.cuda.synchronize()
    assert torch.equal(opt_g.scaler.state, opt_e.scaler.state)
    assert opt_g.scaler.steps_skipped >= 1
    for (n, a), (_, b) in zip(net_g.state_dict().items(), net_e.state_dict().items()):
        assert torch.equal(a, b), n
