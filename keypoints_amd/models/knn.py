"""Building blocks with the reference's names (keypoints/models/knn.py): Container, Unit,
SpatialSoftmax, SpatialLogSoftmax, GaussianLike.  `Unit.forward` runs the whole
pad/conv/BN/activation/pool/upsample stack through libkp_hip.so (unit_engine.py)."""
from pathlib import Path

import torch
from torch import nn as nn

from ..runtime import default_precision, require_device
from ..unit_engine import GradArena, UnitFunction, build_plan, pair_ok
from .functional import gaussian_like_function, spacial_logsoftmax, spacial_softmax, unlazy


class Container(nn.Module):
    """Base class of the model assemblies; owns the weight init (knn.py:12-23)."""

    def _initialize_weights(self):
        # kaiming-normal(fan_out, relu) convs with zero bias, unit-gamma / zero-beta BatchNorm.
        # Drawn in module order from torch's global RNG, i.e. the same stream as the reference.
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)

    def forward(self, *input):
        raise NotImplementedError()

    def save(self, directory):
        raise NotImplementedError()

    def load(self, directory):
        raise NotImplementedError()

    def set_precision(self, precision):
        """'fp32' (strict parity, apex O0), 'bf16' or 'fp16' (mixed precision, fp32 master weights; the O2 role --
        'fp16' is the reference's own type and wants loss scaling: keypoints_amd.amp.scale_loss)."""
        from ..runtime import precision_dtype
        precision_dtype(precision)   # raises KeyError on an unknown name
        for m in self.modules():
            if isinstance(m, Unit):
                m.precision = precision
        return self


class GaussianLike(nn.Module):
    def __init__(self, sigma=0.1):
        super().__init__()
        self.sigma = sigma

    def forward(self, kp, height, width):
        return gaussian_like_function(kp, height, width, self.sigma)


class SpatialSoftmax(nn.Module):
    def forward(self, heatmap, probs=False):
        return spacial_softmax(heatmap, probs)


class SpatialLogSoftmax(nn.Module):
    def forward(self, heatmap, probs=False):
        return spacial_logsoftmax(heatmap, probs)


class Unit(nn.Module):
    """in_block (pad, 3x3 conv, BN, LeakyReLU) -> core -> out_block (1x1 conv, LeakyReLU)."""

    def __init__(self, in_channels, out_channels, core, batch_norm=True):
        super().__init__()
        convs = [m for m in core.modules() if isinstance(m, nn.Conv2d)]
        core_in, core_out = convs[0].in_channels, convs[-1].out_channels

        head = [nn.ReplicationPad2d(1), nn.Conv2d(in_channels, core_in, kernel_size=3, stride=1)]
        if batch_norm:
            head.append(nn.BatchNorm2d(core_in))
        head.append(nn.LeakyReLU(inplace=True))
        self.in_block = nn.Sequential(*head)
        self.core = core
        self.out_block = nn.Sequential(nn.Conv2d(core_out, out_channels, kernel_size=1, stride=1),
                                       nn.LeakyReLU(inplace=True))
        self.precision = default_precision()
        self._plan = None
        self._arena = None
        self.after_backward = None   # optional hook(unit): data-parallel gradient reduction hangs here
        self.layer_done = None       # optional hook(unit, idx): the gradients of plan steps >= idx are final
        self.sync_bn = None          # (process group, world size): BatchNorm batch statistics over all ranks (dist.GradReducer(sync_bn=True))

    # -- HIP execution -------------------------------------------------------------------------
    def plan(self):
        if self._plan is None:
            self._plan = build_plan([self.in_block, self.core, self.out_block])
        return self._plan

    def grad_arena(self):
        params = list(self.parameters())
        if self._arena is None or self._arena.flat.device != params[0].device:
            self._arena = GradArena(params)
        return self._arena

    def _after_backward(self):
        if self.after_backward is not None:
            self.after_backward(self)

    def _layer_done(self, idx):
        if self.layer_done is not None:
            self.layer_done(self, idx)

    def forward(self, x):
        x = unlazy(x)
        require_device(x)
        params = list(self.parameters())
        record = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return UnitFunction.apply(x, self, record, (False, False), None, None, *params)

    def forward_cat(self, x, x2):
        """self(torch.cat((x, x2), dim=1)) without forming the concatenated tensor (KeyNet.forward, keynet.py:29: the decoder reads
        the encoder's z and the Gaussian maps m): the first layer's layout kernel reads both sources, the backward splits the input
        gradient.  Same values as the cat + forward."""
        x, x2 = unlazy(x), unlazy(x2)
        require_device(x, x2)
        params = list(self.parameters())
        record = torch.is_grad_enabled() and (x.requires_grad or x2.requires_grad or any(p.requires_grad for p in params))
        return UnitFunction.apply(x, self, record, (False, False), None, x2, *params)

    def forward_pair(self, xs, x, out_t=False):
        """-> (self(xs) under no_grad, self(x)): the two calls TransporterNet.forward makes per encoder (transporter.py:36-39), as one
        batch with per-half BatchNorm statistics where the library has the pair kernels (unit_engine.pair_ok: bit-identical to the
        two calls), else as the two calls.  out_t: the outputs in the 16-bit NHWC interchange format (see forward_t)."""
        xs, x = unlazy(xs), unlazy(x)
        require_device(xs, x)
        if (torch.is_tensor(xs) and torch.is_tensor(x) and xs.shape == x.shape and x.dim() == 4
                and pair_ok(self, x.size(0), x.size(1), x.size(2), x.size(3), out_t)):
            params = list(self.parameters())
            record = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
            return UnitFunction.apply(x, self, record, (False, bool(out_t)), xs.detach(), None, *params)
        with torch.no_grad():
            out_s = self.forward_t(xs, out_t=out_t) if out_t else self(xs)
        return out_s, (self.forward_t(x, out_t=out_t) if out_t else self(x))

    def forward_t(self, x, in_t=False, out_t=False):
        """`forward` with the 16-bit NHWC interchange formats of include/kp_hip.h at either end (UnitFunction.forward): what
        TransporterNet uses between the feature encoder, the keypoint bottleneck and the decoder where the library has the kernels."""
        x = unlazy(x)
        require_device(x)
        params = list(self.parameters())
        record = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return UnitFunction.apply(x, self, record, (bool(in_t), bool(out_t)), None, None, *params)

    # -- checkpoints: same files / keys as knn.py:143-167 ---------------------------------------
    def _blocks(self, in_block=True, core=True, out_block=True):
        return [(n, getattr(self, n)) for n, on in (('in_block', in_block), ('core', core), ('out_block', out_block))
                if on]

    def save(self, directory):
        Path(directory).mkdir(parents=True, exist_ok=True)
        for name, block in self._blocks():
            # (FusedAdam.bind homes every parameter as a view of ONE per-device buffer and torch.save serialises the whole storage
            #  behind a view: detached copies keep each file at its block's own size, same keys and values)
            torch.save({k: v.detach().clone() for k, v in block.state_dict().items()}, str(Path(directory) / f'{name}.mdl'))

    def load(self, directory, in_block=True, core=True, out_block=True, map_device=None):
        for name, block in self._blocks(in_block, core, out_block):
            path = str(Path(directory) / f'{name}.mdl')
            block.load_state_dict(torch.load(path, map_location=map_device) if map_device else torch.load(path))
