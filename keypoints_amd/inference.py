"""Inference-only keypoint extractor.

The downstream consumers of a trained Transporter run only `net.keypoint(x)` + `spacial_logsoftmax` with BatchNorm in
eval mode, one small batch at a time (atari_demo.py:35-36, tests/test_gl_hack.py:454-456 of the reference).  At that
size the ~60 kernels of a Pong-class keypoint network are launch-bound, so the chain is captured once per input shape
as a HIP graph (`torch.cuda.CUDAGraph` is hipGraph on ROCm; the C-ABI calls enqueue on the capturing stream like any
other) and replayed: one host call per frame instead of one per kernel.

Pong-class keypoint networks (no pooling / upsampling in the core, channel widths that are multiples of 8:
`vgg_cfg['VGG_PONG_LAYERNECK' | 'VGG_PONG_TRIVIAL' | 'VGG_PACMAN' | 'MAPPER']`, keypoints/models/vgg.py:64-70) take the
fused route instead: the eval-mode BatchNorm is folded into the convolution weights once, and the whole Unit runs as ONE
fp32 kernel per frame batch (`kp_unit_infer_fwd`, keypoint_amd/csrc/kp_infer.hip) followed by the spatial log-softmax --
two launches instead of ~30, with or without the graph.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import call, ptr
from .models import functional as MF
from .runtime import stream_ptr
from .unit_engine import shared_packs


class FusedUnit:
    """Eval-mode forward of a small-channel Unit as one kernel (`kp_unit_infer_fwd`).

    Built from a `knn.Unit`; raises NotImplementedError when the Unit is not of the fusable form (a pooling / upsampling
    step, a width that is not a multiple of 8, more than 6 layers, a BatchNorm without running statistics).  The folded
    weights are a snapshot: build a new FusedUnit after changing the network's parameters or buffers."""

    def __init__(self, unit):
        plan = unit.plan()
        convs, head = plan[:-1], plan[-1]
        if any(s['tail'] is not None for s in convs):
            raise NotImplementedError('fused inference: the core has a pooling / upsampling step')
        for s in convs:
            if s['bn'] is not None and (s['bn'].running_mean is None or s['bn'].running_var is None):
                raise NotImplementedError('fused inference: BatchNorm without running statistics')
        widths = [s['conv'].out_channels for s in convs] + [head['conv'].out_channels]
        slopes = [s['slope'] for s in convs] + [head['slope']]
        self.cin = convs[0]['conv'].in_channels
        self.L = len(convs)
        self.K = widths[-1]
        self._widths = (ctypes.c_int * len(widths))(*widths)
        self._slopes = (ctypes.c_float * len(slopes))(*slopes)
        n = int(_lib.lib().kp_unit_infer_param_floats(self.cin, self.L, self._widths))
        if n <= 0:
            raise NotImplementedError(f'fused inference: widths {widths} ({self.L} 3x3 layers) have no fused kernel')
        dev = head['conv'].weight.device
        if dev.type != 'cuda':
            raise RuntimeError('FusedUnit: the Unit must live on a HIP device (no CPU fallback)')
        # fold + reorder on the device, straight from the state_dict tensors (kp_unit_infer_pack, one launch per layer)
        self.params = torch.empty(n, device=dev, dtype=torch.float32)
        st = stream_ptr()
        f32 = lambda t: None if t is None else t.detach().float().contiguous()
        for l, s in enumerate(plan):
            conv, bn = s['conv'], s.get('bn')
            keep = [f32(conv.weight), f32(conv.bias)] + ([f32(bn.weight), f32(bn.bias), f32(bn.running_mean), f32(bn.running_var)]
                                                         if bn is not None else [None] * 4)
            call('kp_unit_infer_pack', ptr(self.params), l, self.cin, self.L, self._widths, *[ptr(t) for t in keep],
                 float(bn.eps) if bn is not None else 0.0, st)
        torch.cuda.current_stream(dev).synchronize()      # `keep` may hold temporaries

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda or x.dim() != 4 or x.shape[1] != self.cin:
            raise RuntimeError(f'FusedUnit: expected a HIP tensor [B, {self.cin}, H, W]')
        x = x.float().contiguous()     # (no-ops for a contiguous float32 frame; the layer chain converts its input the same way)
        B, _, H, W = x.shape
        heat = torch.empty(B, self.K, H, W, device=x.device, dtype=torch.float32)
        call('kp_unit_infer_fwd', ptr(x), ptr(self.params), ptr(heat), B, self.cin, H, W, self.L, self._widths, self._slopes,
             stream_ptr())
        return heat

    def keypoints(self, x: torch.Tensor):
        """`spacial_logsoftmax(unit(x))` as one library call: (k [B, K, 2], heat [B, K, H, W])."""
        if not x.is_cuda or x.dim() != 4 or x.shape[1] != self.cin:
            raise RuntimeError(f'FusedUnit: expected a HIP tensor [B, {self.cin}, H, W]')
        x = x.float().contiguous()     # (no-ops for a contiguous float32 frame; the layer chain converts its input the same way)
        B, _, H, W = x.shape
        K = self.K
        buf = torch.empty(B * K * (H * W + 2 + H + W), device=x.device, dtype=torch.float32)   # one allocation: heat | k | p | q
        n0 = B * K * H * W
        heat, k = buf[:n0].view(B, K, H, W), buf[n0:n0 + 2 * B * K].view(B, K, 2)
        p0 = n0 + 2 * B * K
        call('kp_unit_infer_keypoints_fwd', ptr(x), ptr(self.params), ptr(heat), ptr(k), buf.data_ptr() + 4 * p0,
             buf.data_ptr() + 4 * (p0 + B * K * H), B, self.cin, H, W, self.L, self._widths, self._slopes, stream_ptr())
        return k, heat


class KeypointExtractor:
    """`k = KeypointExtractor(net)(x)`: keypoints `[B, K, 2]` (y, x in [0, 1]) of `x [B, C, H, W]`.

    net: a `TransporterNet` / `KeyNet` (anything with a `.keypoint` Unit); it is switched to eval mode.
    graph=True captures one HIP graph per (shape, dtype) of `x`; the returned tensor is then a static buffer that the
    next call with the same shape overwrites (clone it to keep it).  `heatmap` of the last call is kept on the object.
    The weights are taken as frozen: their packed bf16 copies are made once and kept (not part of the graph); call
    `refresh()` after changing them.
    """

    def __init__(self, net, graph: bool = True, warmup: int = 3, fused='auto'):
        """fused: 'auto' = the one-kernel Unit when the keypoint network has that form, True = require it, False = always
        the layer-by-layer chain (the training kernels in eval mode, in the network's precision)."""
        self.net = net.eval()
        self.graph = graph
        self.warmup = warmup
        self.fused = fused
        self._captured = {}
        self._packs = {}
        self._fused_unit = None
        self._fused_pending = False
        self.heatmap = None
        self._build_fused()

    def _build_fused(self):
        self._fused_unit = None
        self._fused_pending = False
        if self.fused is False:
            return
        if next(self.net.keypoint.parameters()).device.type != 'cuda':
            self._fused_pending = True     # the network is moved to the device later: fold at the first call
            return
        try:
            self._fused_unit = FusedUnit(self.net.keypoint)
        except NotImplementedError:
            if self.fused is True:
                raise

    def refresh(self):
        """Forget the packed / folded weights and the captured graphs (the network's parameters changed)."""
        self._captured.clear()
        self._packs.clear()
        self._build_fused()

    @torch.no_grad()
    def _run(self, x):
        if self._fused_pending:
            self._build_fused()
        if self._fused_unit is not None:
            return self._fused_unit.keypoints(x)
        else:
            with shared_packs(self._packs):
                heat = self.net.keypoint(x)
        return MF.spacial_logsoftmax(heat), heat

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError('KeypointExtractor runs on a HIP device only (no CPU fallback)')
        if not self.graph:
            k, self.heatmap = self._run(x)
            return k
        key = (tuple(x.shape), x.dtype, x.device.index)
        ent = self._captured.get(key)
        if ent is None:
            static_in = x.clone()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):          # warm-up off the capture: lazy initialisation, allocator pools
                for _ in range(self.warmup):
                    self._run(static_in)
            torch.cuda.current_stream(x.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                k, heat = self._run(static_in)
            ent = self._captured[key] = (g, static_in, k, heat)
        g, static_in, k, heat = ent
        static_in.copy_(x)
        g.replay()
        self.heatmap = heat
        return k
