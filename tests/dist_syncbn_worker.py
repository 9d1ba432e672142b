"""Worker of tests/test_gpu_dist_nccl.py::test_sync_bn_equals_the_concatenated_batch (run under torch.distributed.run,
gloo, the ranks share cuda:0): one fp32 train step of a Pong-class Transporter on this rank's half of a batch with
GradReducer(sync_bn=True); rank 0 also runs the whole batch in one process and reports the differences."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import keypoints_amd as ka
from keypoints_amd.dist import GradReducer
from keypoints_amd.models import transporter

ka.set_multi_stream(False)   # both ranks share the test box's one GPU: side streams of two processes oversubscribe its queues


def make(seed):
    torch.manual_seed(seed)
    typ = os.environ.get('KP_SYNCBN_TYPE', 'VGG_PONG_LAYERNECK')
    cin = 1 if typ.startswith('VGG_PONG') else 3
    return transporter.make(typ, cin, 16, 4).cuda().set_precision('fp32'), cin


def step(net, xs, xt, reducer=None):
    opt = ka.FusedAdam(net.parameters(), lr=1e-3).bind(net)
    opt.zero_grad()
    out = net(xs, xt)
    loss = ka.l2_reconstruction_loss(out[0], xt)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    opt.step()
    return float(loss.detach()), grads


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    size, B = int(os.environ.get('KP_SYNCBN_SIZE', 32)), 8
    net, cin = make(0)
    g = torch.Generator().manual_seed(5)
    xs = torch.rand(B, cin, size, size, generator=g).cuda()
    xt = torch.rand(B, cin, size, size, generator=g).cuda()
    lo, hi = rank * B // world, (rank + 1) * B // world
    reducer = GradReducer(net, sync_bn=True, overlap=False)
    loss, grads = step(net, xs[lo:hi], xt[lo:hi], reducer)
    losses = torch.tensor([loss], dtype=torch.float64)
    dist.all_reduce(losses)
    reducer.detach()
    if rank == 0:
        ref, _ = make(0)
        rloss, rgrads = step(ref, xs, xt)

        def rel(a, b):
            return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
        worst = max((rel(grads[n], rgrads[n]), n) for n in grads if float(rgrads[n].norm()) > 1e-12)
        pw = max(rel(p, q) for p, q in zip(net.parameters(), ref.parameters()))
        bw = max(rel(a, b) for (_, a), (_, b) in zip(net.named_buffers(), ref.named_buffers()) if a.dtype.is_floating_point)
        print('SYNCBN ' + json.dumps(dict(loss_dp=float(losses) / world, loss_ref=rloss, worst_grad=worst[0], worst_name=worst[1],
                                          params=pw, buffers=bw)), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
