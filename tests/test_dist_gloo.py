"""CPU, world_size 2 over gloo: the data-parallel gradient reduction used for N > 1 GPUs
(keypoints_amd/dist.py).  On the GPU box the same code runs over RCCL (backend 'nccl')."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from keypoints_amd.dist import GradReducer
        from keypoints_amd.models import transporter
        torch.manual_seed(100 + rank)                      # deliberately different init per rank
        net = transporter.make('VGG_PONG_LAYERNECK', 1, 16, 3)
        red = GradReducer(net)                              # broadcasts rank 0's parameters + buffers
        w0 = net.feature.in_block[1].weight.detach().clone()
        ref = [torch.empty_like(w0) for _ in range(world)]
        dist.all_gather(ref, w0)
        same_init = all(torch.equal(ref[0], r) for r in ref)
        seen = red.ranks_seen()      # bench.py's dp.nccl_ranks: a SUM all-reduce of ones through the gradient transport itself
        same_init = same_init and seen['allreduce_of_ones'] == world and seen['comm_count'] == world and 'gloo' in seen['transport']
        # emulate the end of each Unit's backward: arenas hold rank-dependent gradients
        units = [net.decoder, net.keypoint, net.feature]    # backward order
        pieces = 0
        for i, u in enumerate(units):
            arena = u.grad_arena()
            arena.attach(list(u.parameters()))
            arena.flat.copy_(torch.arange(arena.flat.numel(), dtype=torch.float32) * (rank + 1) + i)
            for idx in range(len(u.plan()) - 2, -1, -1):    # the 3x3 layers report in, deepest first ...
                u._layer_done(idx)                          # ... the reducer sends the arena's tail at its split
            u._after_backward()                             # and the rest at the end of the Unit's backward
            pieces += 2 if red._split[id(u)] is not None else 1
        split_ok = len(red._pending) == pieces and any(red._split[id(u)] is not None for u in units)
        for u in units:                                     # the tail piece covers >= 60 % of the arena
            sp = red._split[id(u)]
            if sp is not None:
                n = u.grad_arena().flat.numel()
                split_ok = split_ok and 0 < sp[1] <= 0.4 * n
        red.finish()
        ok = split_ok
        for i, u in enumerate(units):
            n = u.grad_arena().flat.numel()
            expect = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world) + i
            ok = ok and torch.allclose(u.grad_arena().flat, expect, rtol=1e-6, atol=1e-6)
            ok = ok and all(p.grad is u.grad_arena().views[id(p)] for p in u.parameters())
        pending_after = len(red._pending)

        # (2) a second backward into a Unit whose arena is in flight must raise, not corrupt the gradients
        u = units[0]
        for idx in range(len(u.plan()) - 2, -1, -1):
            u._layer_done(idx)
        u._after_backward()
        raised = False
        try:
            for idx in range(len(u.plan()) - 2, -1, -1):
                u._layer_done(idx)
            u._after_backward()
        except RuntimeError as e:
            raised = 'no_sync' in str(e)
        red.finish()

        # (3) micro-batch accumulation: backwards under no_sync() send nothing; the last one reduces the sum ONCE
        acc_ok = True
        arena = u.grad_arena()
        arena.flat.fill_(float(rank + 1))
        with red.no_sync():
            for idx in range(len(u.plan()) - 2, -1, -1):
                u._layer_done(idx)
            u._after_backward()
            acc_ok = acc_ok and red.pieces_issued == 0
        arena.flat.add_(float(rank + 1))                    # the second micro-batch accumulates
        for idx in range(len(u.plan()) - 2, -1, -1):
            u._layer_done(idx)
        u._after_backward()
        red.finish()
        mean2 = 2.0 * sum(range(1, world + 1)) / world
        acc_ok = acc_ok and torch.allclose(arena.flat, torch.full_like(arena.flat, mean2))

        # (4) bf16 transport: same mean within bf16 rounding of each rank's contribution
        red16 = GradReducer(net, broadcast_init=False, transport='bf16')
        g0 = torch.linspace(-1, 1, arena.flat.numel()) * (rank + 1)
        arena.flat.copy_(g0)
        for idx in range(len(u.plan()) - 2, -1, -1):
            u._layer_done(idx)
        u._after_backward()
        red16.finish()
        expect16 = torch.linspace(-1, 1, arena.flat.numel()) * (sum(range(1, world + 1)) / world)
        bf_ok = float((arena.flat - expect16).abs().max()) <= 2 ** -7 and arena.flat.dtype == torch.float32

        # (4b) overlap=False: the hooks send nothing, finish() reduces the same pieces back to back
        red_late = GradReducer(net, broadcast_init=False, overlap=False)
        arena.flat.fill_(float(rank + 1))
        for idx in range(len(u.plan()) - 2, -1, -1):
            u._layer_done(idx)
        u._after_backward()
        late_ok = len(red_late._pending) == 0 and len(red_late._ready) == red_late.pieces_issued == 1   # the whole arena, once
        red_late.finish()
        late_ok = late_ok and torch.allclose(arena.flat, torch.full_like(arena.flat, sum(range(1, world + 1)) / world))
        late_ok = late_ok and not red_late._ready and not red_late._pending

        # (4c) overlap=False with every Unit's arena a slice of ONE buffer (what FusedAdam.bind sets up on the device): finish()
        # sends the whole network as ONE collective; Units with arenas of their own still go out one by one
        from keypoints_amd.unit_engine import GradArena
        sizes = [sum(p.numel() for p in u.parameters()) for u in units]
        starts, tot = [], 0
        for n in sizes:
            starts.append(tot); tot += (n + 63) // 64 * 64
        g_all = torch.zeros(tot)
        for u, s0, n in zip(units, starts, sizes):
            u._arena = GradArena(list(u.parameters()), flat=g_all[s0:s0 + n])
        red_one = GradReducer(net, broadcast_init=False, overlap=False)
        for i, u in enumerate(units):
            u.grad_arena().flat.fill_(float(rank + 1) + i)
            for idx in range(len(u.plan()) - 2, -1, -1):
                u._layer_done(idx)
            u._after_backward()
        one_ok = len(red_one._ready) == 3
        red_one.finish()
        one_ok = one_ok and red_one.collectives_last_finish == 1
        for i, u in enumerate(units):
            want = sum(range(1, world + 1)) / world + i
            one_ok = one_ok and torch.allclose(u.grad_arena().flat, torch.full_like(u.grad_arena().flat, want))
        gaps = torch.ones(tot, dtype=torch.bool)
        for s0, n in zip(starts, sizes):
            gaps[s0:s0 + n] = False
        one_ok = one_ok and float(g_all[gaps].abs().sum()) == 0.0       # the padding between the slices stays zero
        units[1]._arena = None                                           # a Unit with an arena of its own again
        units[1].grad_arena()
        red_mix = GradReducer(net, broadcast_init=False, overlap=False)
        for u in units:
            u.grad_arena().flat.fill_(float(rank + 1))
            for idx in range(len(u.plan()) - 2, -1, -1):
                u._layer_done(idx)
            u._after_backward()
        red_mix.finish()
        one_ok = one_ok and red_mix.collectives_last_finish == 3         # (the two slices are no longer neighbours: gap of a whole arena)
        one_ok = one_ok and all(torch.allclose(u.grad_arena().flat, torch.full_like(u.grad_arena().flat, sum(range(1, world + 1)) / world))
                                for u in units)
        late_ok = late_ok and one_ok

        # (5) checkpoint: per-rank BatchNorm buffers -> rank 0's on every rank, rank 0 writes the reference's layout
        import tempfile
        for b in net.buffers():
            if b.dtype.is_floating_point:
                b.fill_(float(rank + 3))
        d = tempfile.mkdtemp(prefix=f'kp_dp_{rank}_')
        red16.save(d)
        buf_ok = all(float(b.float().mean()) == 3.0 for b in net.buffers() if b.dtype.is_floating_point)
        wrote = os.path.exists(os.path.join(d, 'encoder', 'core.mdl'))
        buf_ok = buf_ok and (wrote == (rank == 0))
        q.put((rank, same_init, ok and raised and acc_ok and bf_ok and buf_ok and late_ok, pending_after,
               dict(base=ok, raised=raised, acc=acc_ok, bf16=bf_ok, save=buf_ok, late=late_ok)))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_init, ok, pending, detail in res:
        assert same_init, f'rank {rank}: parameters not broadcast from rank 0'
        assert ok, f'rank {rank}: {detail}'
        assert pending == 0
