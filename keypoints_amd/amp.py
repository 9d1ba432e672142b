"""Loss scaling for the fp16 storage mode: what `apex.amp` does for the reference's train step
(transporter.py:51 `amp.initialize(net, optim, opt_level=...)`, :84-86 `with amp.scale_loss(loss, optim) as scaled_loss:
scaled_loss.backward()`), with the scale, the overflow flag, Adam's step count and its bias corrections in a small
device-resident state block (include/kp_hip.h, kp_amp_*): no host synchronisation per step.

    net, optim = amp.initialize(net, optim, opt_level='O2')          # 'O0' fp32, 'O1'/'O2' fp16 + dynamic loss scale
    with amp.scale_loss(loss, optim) as scaled_loss:
        scaled_loss.backward()
    optim.step()                                                     # checks, un-scales, updates or skips, adapts

bf16 storage (`set_precision('bf16')`) has fp32's exponent range and needs none of this.
"""
from __future__ import annotations

import contextlib

import torch

from ._lib import call, ptr
from .runtime import f32c, require_device, stream_ptr


class LossScaler:
    """Device-resident dynamic loss scale (defaults = apex's dynamic scaler: 2^16, x2 every 2000 clean steps, /2 on overflow)."""

    def __init__(self, device, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000,
                 max_scale=2.0 ** 24):
        self.state = torch.zeros(8, device=device, dtype=torch.float64)
        self.growth_factor, self.backoff_factor = float(growth_factor), float(backoff_factor)
        self.growth_interval, self.max_scale = int(growth_interval), float(max_scale)
        call('kp_amp_init', ptr(self.state), float(init_scale), stream_ptr())

    # host-side views for logging / tests (each is a device read: not for the train loop)
    @property
    def scale(self) -> float:
        return float(self.state[0].item())

    @property
    def steps_skipped(self) -> int:
        return int(self.state[6].item())

    @property
    def adam_steps(self) -> int:
        return int(self.state[3].item())

    def scale_tensor(self) -> torch.Tensor:
        """fp32 0-dim device tensor holding the current scale (a kernel argument, never read on the host)."""
        return self.state[0:1].to(torch.float32)


class _ScaleLoss(torch.autograd.Function):
    """loss * scale with the scale read from the scaler's device state block (never on the host): ONE kp_amp_scale launch each
    way -- no fp64 -> fp32 copy of the scale, no torch multiplication between the loss kernel and the backward's first kernel."""

    @staticmethod
    def forward(ctx, loss, state):
        require_device(loss, state)
        ctx.state = state          # (the scaler's own buffer: it outlives the step)
        x = f32c(loss).reshape(1)
        y = torch.empty_like(x)
        call('kp_amp_scale', ptr(x), ptr(state), ptr(y), 1, stream_ptr())
        return y.reshape(())

    @staticmethod
    def backward(ctx, g):
        x = f32c(g).reshape(1)
        y = torch.empty_like(x)
        call('kp_amp_scale', ptr(x), ptr(ctx.state), ptr(y), 1, stream_ptr())
        return y.reshape(()), None


def _require_scaling_optimizer(optimizer, who):
    """A loss scaler only works with an optimizer whose step() un-scales the gradients, skips non-finite steps and adapts the
    scale (keypoints_amd.FusedAdam).  On any other optimizer -- torch.optim.Adam, which is what transporter.py:47 builds --
    `.scaler` would be an ignored attribute: the loss multiplied by 2^16 and nothing dividing the gradients again."""
    from .optim import FusedAdam
    if not (isinstance(optimizer, FusedAdam) or getattr(optimizer, 'honours_loss_scaler', False)):
        raise TypeError(f'{who}: {type(optimizer).__name__} does not un-scale gradients; use keypoints_amd.FusedAdam '
                        '(or an optimizer that sets `honours_loss_scaler = True` and divides by `.scaler` in step())')


def initialize(net, optimizer, opt_level='O2', loss_scale='dynamic', **scaler_kwargs):
    """apex.amp.initialize's role: 'O0' -> strict fp32 Units; 'O1' / 'O2' / 'O3' -> fp16 Units with fp32 master
    weights and (loss_scale='dynamic') a device-resident dynamic loss scaler attached to `optimizer`."""
    if opt_level == 'O0':
        net.set_precision('fp32')
        return net, optimizer
    _require_scaling_optimizer(optimizer, 'amp.initialize')
    net.set_precision('fp16')
    dev = next(net.parameters()).device
    if loss_scale == 'dynamic':
        optimizer.scaler = LossScaler(dev, **scaler_kwargs)
    else:
        optimizer.scaler = LossScaler(dev, init_scale=float(loss_scale), growth_factor=1.0, backoff_factor=1.0,
                                      growth_interval=1 << 30)
    return net, optimizer


@contextlib.contextmanager
def scale_loss(loss, optimizer):
    """`with amp.scale_loss(loss, optim) as scaled_loss: scaled_loss.backward()` (transporter.py:84-86).  Without a
    scaler on the optimizer (fp32 / bf16 modes) the loss passes through unchanged."""
    scaler = getattr(optimizer, 'scaler', None)
    if scaler is None or getattr(scaler, 'unit', False):   # (unit: constant scale 1, keypoints_amd.graph)
        yield loss
    else:
        yield _ScaleLoss.apply(loss, scaler.state)
