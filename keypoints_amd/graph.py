"""Whole-train-step HIP graph: `zero_grad -> forward -> loss -> backward -> optimiser step` captured once and replayed.

At the reference's shipped batch sizes (16-32, configs/*.yaml:14-18) a step of model F is ~400 kernel launches of a few
microseconds each and the Python / launch path, not the GPU, sets the pace; nothing in the step synchronises with the host
(statistics, loss scale, Adam's step count and bias corrections all live on the device), so the whole step can be one graph
launch.  The side streams of the step (runtime.fork / side_stream) become parallel branches of the graph.

    step = keypoints_amd.graph.GraphedTrainStep(net, optim, loss_fn, (xs, xt))     # warm-up steps (these TRAIN) + capture (runs nothing)
    for xs, xt in loader:
        loss = step(xs, xt)          # copies the batch into the graph's input buffers, replays, returns the loss tensor
                                     # (a static buffer: the next call overwrites it -- clone() to keep a value)

Frozen at capture: input shapes, the precision mode, every hyper-parameter that reaches a kernel as a scalar argument
(learning rate, betas, eps) -- build a new GraphedTrainStep to change one.  Not for the data-parallel reducer (its
collectives are issued from host hooks).
"""
from __future__ import annotations

import torch

from . import amp, runtime
from .optim import l2_reconstruction_loss


def _default_loss(out, inputs):
    return l2_reconstruction_loss(out[0], inputs[-1])


class GraphedTrainStep:
    def __init__(self, net, optimizer, loss_fn=None, example_inputs=(), warmup: int = 3):
        """loss_fn(outputs, inputs) -> scalar loss; default: the reference's criterion on (out[0], last input)
        (transporter.py:80-83: `x_t` against `xt`)."""
        self.net, self.opt = net, optimizer
        self.loss_fn = loss_fn or _default_loss
        if not example_inputs:
            raise ValueError('GraphedTrainStep needs example inputs (their shapes are frozen into the graph)')
        dev = example_inputs[0].device
        amp._require_scaling_optimizer(optimizer, 'GraphedTrainStep')
        self._borrowed_scaler = getattr(optimizer, 'scaler', None) is None
        if optimizer.scaler is None:
            # Adam's step count and bias corrections must advance ON THE DEVICE inside a replayed graph: use the loss-scaler
            # state block with a constant scale of 1 (bf16 / fp32 modes: nothing is scaled, a non-finite gradient skips the
            # step instead of poisoning the weights)
            optimizer.scaler = amp.LossScaler(dev, init_scale=1.0, growth_factor=1.0, backoff_factor=1.0,
                                              growth_interval=1 << 30, max_scale=1.0)
            optimizer.scaler.unit = True
        self._in = [x.detach().clone() for x in example_inputs]
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):   # allocator pools, side streams, autotuned plans: all created before the capture
                self._step()
        cur.wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # capture on the SAME stream the warm-up ran on: runtime.side_stream() keys the side streams by their parent stream, so
        # the 'chain' / 'wgrad' streams (and their allocator pools) the capture forks into are the warmed ones, not streams
        # created for the first time inside the capture
        with torch.cuda.graph(self.graph, stream=s):
            self.loss = self._step()

    def release(self):
        """Give the optimizer back as it was: the unit loss scaler this object attached (bf16 / fp32 modes) stays on the
        optimizer for as long as the graph may be replayed -- the replayed Adam kernels read its device state block -- and
        would make later EAGER steps pay a kp_amp_check per arena and skip non-finite steps silently.  Call when the graph is
        discarded and training continues eagerly."""
        if self._borrowed_scaler and getattr(self.opt.scaler, 'unit', False):
            self.opt.scaler = None
        self.graph = None

    def _step(self):
        self.opt.zero_grad()
        out = self.net(*self._in)
        loss = self.loss_fn(out, self._in)
        with amp.scale_loss(loss, self.opt) as scaled:
            scaled.backward()
        runtime.join_side_streams()
        self.opt.step()
        return loss.detach()

    def __call__(self, *inputs):
        for dst, src in zip(self._in, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.loss
