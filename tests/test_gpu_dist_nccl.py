"""The data-parallel reducer over RCCL (torch.distributed backend 'nccl') on the one GPU a test box has: world size 1,
but the real code path -- RCCL communicator, async all-reduce pieces on RCCL's stream launched from the backward hooks of a
real Transporter backward, finish() joining them before Adam (keypoints_amd/dist.py).  World-size-2 semantics (averaging,
piece bookkeeping, bf16 transport, no_sync, checkpoint broadcast) are covered on CPU over gloo (tests/test_dist_gloo.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.fixture()
def nccl_world1():
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    yield
    dist.destroy_process_group()


def _step(net, opt, xs, xt, reducer=None):
    import keypoints_amd as ka
    opt.zero_grad()
    loss = ka.l2_reconstruction_loss(net(xs, xt)[0], xt)
    loss.backward()
    issued = None
    if reducer is not None:
        issued = reducer.pieces_issued
        reducer.finish()
    opt.step()
    return loss, issued


@pytest.mark.parametrize('transport', ['fp32', 'bf16'])
def test_grad_reducer_nccl_world1_real_backward(nccl_world1, transport):
    import keypoints_amd as ka
    from keypoints_amd.dist import GradReducer
    from keypoints_amd.models import transporter
    g = torch.Generator().manual_seed(3)
    xs = torch.rand(8, 3, 128, 128, generator=g).cuda()
    xt = torch.rand(8, 3, 128, 128, generator=g).cuda()
    nets = []
    for with_dp in (False, True):
        torch.manual_seed(11)
        net = transporter.make('F', 3, 64, 10).cuda().set_precision('bf16')
        opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
        # (reserve_cus=0: the persistent kernels keep the whole device, so that the DP step can be compared BIT FOR BIT with the
        #  non-DP one; the reduced grid of the default overlap mode has its own test below)
        red = GradReducer(net, transport=transport, reserve_cus=0) if with_dp else None
        if red is not None:
            assert all(red._split[id(u)] is not None for u in red.units)      # model F: every Unit sends its tail early
        for _ in range(2):
            loss, issued = _step(net, opt, xs, xt, red)
        if red is not None:
            assert issued == 2 * len(red.units), issued                        # two pieces per Unit, all before finish()
            assert not red._pending and red.pieces_issued == 0
            # a second backward into in-flight arenas is refused ...
            opt.zero_grad()
            ka.l2_reconstruction_loss(net(xs, xt)[0], xt).backward()
            with pytest.raises(RuntimeError, match='no_sync'):
                ka.l2_reconstruction_loss(net(xs, xt)[0], xt).backward()
            for work, _, _ in red._pending:
                work.wait()
            torch.cuda.synchronize()
            red._pending.clear(); red._early.clear(); red._done.clear(); red.pieces_issued = 0
            red.detach()
        nets.append((net, float(loss)))
    (ref, l0), (dp, l1) = nets
    if transport == 'fp32':     # AVG over one rank is the identity and the step is bitwise reproducible: identical parameters
        assert abs(l0 - l1) <= 1e-6 * abs(l0)
        for (n, a), (_, b) in zip(ref.named_parameters(), dp.named_parameters()):
            assert torch.equal(a, b), n
    else:                       # gradients went through bf16 on the wire: 2^-9 relative per element
        assert abs(l0 - l1) <= 1e-3 * abs(l0)
        for (n, a), (_, b) in zip(ref.named_parameters(), dp.named_parameters()):
            assert float((a - b).abs().max()) <= 2.01e-4, n          # Adam steps of lr = 1e-4 in the worst case flip sign


@pytest.mark.parametrize('overlap', [0, 1])
def test_bench_two_ranks_share_the_gpu_over_gloo(overlap):
    """The driver's N > 1 launch line (torch.distributed.run, one process per rank, barrier + max-over-ranks timing, rank 0
    prints the JSON line) on the one GPU of a test box: two ranks share cuda:0 and exchange gradients over gloo
    (`bench.py --backend gloo`; the reported number is not a scaling result).  Everything but the RCCL transport is the path
    of `bench.py --gpus 2`: sharded batch, GradReducer hooks (both modes), Adam on identical averaged gradients."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--batch', '16', '--backend', 'gloo', '--dp-overlap', str(overlap), '--profile-steps', '0']
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 alone reports
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['global_batch'] == 32 and rec['config']['parallelism'] == 'dp2'
    assert rec['config']['dp_overlap'] is bool(overlap) and rec['scaling'] == 'weak'
    import math
    assert rec['value'] > 0 and rec['config']['loss'] > 0 and math.isfinite(rec['config']['loss']) and 'cpu_baseline' not in rec
    # the self-check of the multi-GPU record: the collective itself counted two ranks, the exposed exchange was timed, and both
    # placements of the all-reduces ran back to back
    dp = rec['dp']
    assert dp['nccl_ranks']['allreduce_of_ones'] == 2 and dp['nccl_ranks']['comm_count'] == 2
    assert dp['exposed_allreduce_ms_median'] is not None and dp['exposed_allreduce_ms_median'] >= 0
    assert set(dp['modes']) == {'dp_overlap_0', 'dp_overlap_1'} and all(m['value'] > 0 for m in dp['modes'].values())
    assert dp['gradient_bytes_per_step'] == 4 * 24468813      # model F: 97.9 MB of fp32 gradients


@pytest.mark.parametrize('typ,size', [('VGG_PONG_LAYERNECK', 32), ('VGG_PACMAN_2', 32)])
def test_sync_bn_equals_the_concatenated_batch(typ, size):
    """GradReducer(sync_bn=True): two ranks (sharing the GPU, gloo) each run half of a batch; with the BatchNorm
    statistics and backward sums all-reduced per layer, loss, averaged gradients, updated parameters and running
    statistics equal those of ONE process on the whole batch (fp32: to rounding)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', KP_SYNCBN_TYPE=typ, KP_SYNCBN_SIZE=str(size))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'tests', 'dist_syncbn_worker.py')]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('SYNCBN ')]
    assert len(line) == 1, out.stdout[-2000:]
    rec = json.loads(line[0][7:])
    assert abs(rec['loss_dp'] - rec['loss_ref']) <= 2e-6 * abs(rec['loss_ref']), rec
    assert rec['worst_grad'] <= 2e-4, rec       # fp32 summation order differs between the two runs
    assert rec['params'] <= 1e-5 and rec['buffers'] <= 1e-5, rec


def test_native_comm_world1(nccl_world1):
    """kp_comm_* (the C ABI's own RCCL communicator) on the one GPU of the box: unique id, init, in-place
    all-reduce (average / sum, fp32 and bf16), broadcast, destroy -- and the reducer on it through a real backward:
    with one rank the averaged gradient is the gradient, so the step equals the non-DP step bit for bit."""
    import keypoints_amd as ka
    from keypoints_amd import _lib
    from keypoints_amd.dist import GradReducer, NativeComm
    from keypoints_amd.models import transporter
    assert _lib.lib().kp_comm_available() == 1
    comm = NativeComm()
    for dt in (torch.float32, torch.bfloat16):
        t = torch.randn(1 << 20, device='cuda').to(dt)
        ref = t.clone()
        comm.all_reduce(t, average=True).wait()
        comm.all_reduce(t, average=False).wait()
        comm.broadcast(t, 0)
        torch.cuda.synchronize()
        assert torch.equal(t, ref)
    assert comm.count() == 1        # ncclCommCount of the library's communicator (kp_comm_count)
    comm.close()

    g = torch.Generator().manual_seed(3)
    xs = torch.rand(8, 3, 128, 128, generator=g).cuda()
    xt = torch.rand(8, 3, 128, 128, generator=g).cuda()
    nets = []
    for mode in (None, 'overlap', 'after'):
        torch.manual_seed(11)
        net = transporter.make('F', 3, 64, 10).cuda().set_precision('bf16')
        opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
        red = GradReducer(net, comm='native', overlap=(mode == 'overlap'), reserve_cus=0) if mode else None   # (full grid: bitwise comparison)
        for _ in range(2):
            loss, issued = _step(net, opt, xs, xt, red)
        if red is not None:
            assert issued == (2 if mode == 'overlap' else 1) * len(red.units)
            torch.cuda.synchronize()
            assert red.ranks_seen() == dict(allreduce_of_ones=1, comm_count=1, transport='kp_comm (RCCL via the C ABI)')
            red.detach()
        nets.append(net)
    for other in nets[1:]:
        for (n, a), (_, b) in zip(nets[0].named_parameters(), other.named_parameters()):
            assert torch.equal(a, b), n


def test_overlap_mode_runs_the_persistent_kernels_on_a_fixed_reduced_grid(nccl_world1):
    """GradReducer(overlap=True) leaves `reserve_cus` CUs to RCCL: the persistent kernels size their grids to a FIXED budget
    (kp_config(KP_CFG_CUS) = device CUs - 32, a multiple of the 8 XCDs), restored by overlap=False / detach().  With the reduced grid
    every convolution output and data gradient is bit-identical to the full grid's (a tile's arithmetic does not depend on which
    workgroup runs it); BatchNorm statistics and weight gradients are sums over another fixed partition: equal up to the order of
    fp32 additions; the step is still bitwise reproducible on the reduced grid; one whole step agrees with the full grid's to
    storage rounding."""
    import keypoints_amd as ka
    from keypoints_amd import _lib
    from keypoints_amd._lib import call, ptr
    from keypoints_amd.dist import GradReducer
    from keypoints_amd.models import transporter
    from keypoints_amd.runtime import precision_dtype, stream_ptr
    L = _lib.lib()
    total = torch.cuda.get_device_properties(0).multi_processor_count
    assert L.kp_device_cus() == total
    torch.manual_seed(5)
    net = transporter.make('F', 3, 64, 10).cuda().set_precision('fp16')
    # a lone rank has no collective to make room for: the process-global knob is left alone ...
    lone = GradReducer(net, overlap=True)
    assert lone.grid_cus is None and L.kp_device_cus() == total
    lone.detach()
    # ... unless the single-GPU measurement of the reduced grid asks for it (what world > 1 does by default); a budget the caller
    # had set before (bench.py --grid-cus) is what comes back when the reservation ends, not 0
    call('kp_config', _lib.KP_CFG_CUS, total - 8)
    red = GradReducer(net, overlap=True, reserve_alone=True)
    want = (total - 32) // 8 * 8
    assert red.grid_cus == want and L.kp_device_cus() == want
    red.overlap = False
    assert red.grid_cus is None and L.kp_device_cus() == total - 8
    call('kp_config', _lib.KP_CFG_CUS, 0)
    red.overlap = True
    assert L.kp_device_cus() == want
    red.detach()
    assert L.kp_device_cus() == total

    # ---- op level, through the C ABI: one layer of model F at the bench's shape, full grid vs reduced grid
    tdt, code = precision_dtype('fp16')
    N, H, W, C, K = 32, 64, 64, 128, 256
    g = torch.Generator().manual_seed(9)
    a = torch.randn(N, H + 2, W + 2, C, generator=g).cuda().to(tdt)
    wf = (torch.randn(K, 3, 3, C, generator=g) * 0.05).cuda().to(tdt)
    wd = (torch.randn(C, 3, 3, K, generator=g) * 0.05).cuda().to(tdt)
    bias = torch.randn(K, generator=g).cuda()
    dy = torch.zeros(N, H + 4, W + 4, K, device='cuda', dtype=tdt)
    dy[:, 2:-2, 2:-2] = torch.randn(N, H, W, K, generator=g).cuda().to(tdt)
    st = stream_ptr()
    res = {}
    for cus in (0, want):
        call('kp_config', _lib.KP_CFG_CUS, cus)
        try:
            y = torch.full((N, H, W, K), float('nan'), device='cuda', dtype=tdt)
            rows = L.kp_conv_fwd_stats_rows(N, H + 2, W + 2, C, K, 3, code)
            stats = torch.zeros((N * H * W + 63) // 64, 2, K, device='cuda')
            call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, code, st)
            mean, invstd = torch.empty(K, device='cuda'), torch.empty(K, device='cuda')
            ws = torch.empty(128 * K, device='cuda', dtype=torch.float64)
            call('kp_bn_finalize', ptr(stats), rows, ptr(ws), ptr(mean), ptr(invstd), None, None, None, 0.1, 1e-5, N * H * W, K, st)
            gx = torch.full((N, H + 2, W + 2, C), float('nan'), device='cuda', dtype=tdt)
            call('kp_conv_dgrad', ptr(dy), ptr(wd), ptr(gx), N, H, W, K, C, code, st)
            parts = L.kp_conv_wgrad_partials(N, H, W, C, K, 3, 2, code)
            scratch = (torch.zeros if parts == 1 else torch.empty)(parts, K, 3, 3, C, device='cuda')
            call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, W, C, K, 3, 2, code, st)
            dw = torch.zeros(K, C, 3, 3, device='cuda')
            call('kp_wgrad_unpack', ptr(scratch), parts, ptr(dw), K, C, C, 3, st)
            torch.cuda.synchronize()
            res[cus] = (y, gx, mean, invstd, dw, rows)
        finally:
            call('kp_config', _lib.KP_CFG_CUS, 0)
    (y0, gx0, m0, i0, dw0, r0), (y1, gx1, m1, i1, dw1, r1) = res[0], res[want]
    assert torch.equal(y0, y1) and torch.equal(gx0, gx1)                       # data: bit for bit
    assert float((m0 - m1).abs().max()) <= 1e-6 * float(m0.abs().max()) + 1e-7   # statistics: summation order
    assert float((i0 / i1 - 1).abs().max()) <= 1e-6
    assert float((dw0 - dw1).abs().max()) <= 2e-6 * float(dw0.abs().max())

    # ---- one whole train step on the reduced grid: reproducible bit for bit, and equal to the full grid's to storage rounding
    g = torch.Generator().manual_seed(3)
    xs = torch.rand(16, 3, 128, 128, generator=g).cuda()
    xt = torch.rand(16, 3, 128, 128, generator=g).cuda()
    outs = []
    for cus in (0, want, want):
        call('kp_config', _lib.KP_CFG_CUS, cus)
        try:
            torch.manual_seed(11)
            net = transporter.make('F', 3, 64, 10).cuda().set_precision('fp16')
            opt = ka.FusedAdam(net.parameters(), lr=1e-4).bind(net)
            opt.zero_grad()
            out = net(xs, xt)
            loss = ka.l2_reconstruction_loss(out[0], xt)
            loss.backward()
            opt.step()
            torch.cuda.synchronize()
            outs.append((float(loss), out[0].detach().clone(), [p.detach().clone() for p in net.parameters()]))
        finally:
            call('kp_config', _lib.KP_CFG_CUS, 0)
    (l0, x0, p0), (l1, x1, p1), (l2, x2, p2) = outs
    assert l1 == l2 and torch.equal(x1, x2) and all(torch.equal(a_, b_) for a_, b_ in zip(p1, p2))    # reduced grid: reproducible
    assert abs(l0 - l1) <= 1e-3 * abs(l0)      # (measured 1.1e-4: a statistic's last bit flips single bf16 roundings, which the random-init network amplifies)
    assert float((x0 - x1).norm() / x0.norm()) <= 2e-2     # (16-bit storage at random init: a statistic's last bit flips single roundings, 25 BatchNorm layers amplify them)
