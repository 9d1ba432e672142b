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


class FusedUnitF:
    """Eval-mode forward of a matrix-core Unit (model F's class: 3x3 layers with channel widths that are multiples of 64 on maps that
    are multiples of 16, 16-bit storage) with every BatchNorm folded into the epilogue of its convolution (`kp_conv_fwd_act`,
    include/kp_hip.h): per layer ONE matrix-core launch that reads the previous layer's padded activation and writes the next one's --
    no statistics, no BatchNorm pass, no conv output in memory, nothing saved.  What remains beside the convolutions: the
    replicate ring of a padded output (`kp_pad_border`, a few KB per image), the MaxPool2d / UpsamplingBilinear2d steps (the
    training pass kernel with identity constants) and the 1x1 head.  The image-sized first layer (Cin <= 3 -> 64) runs on the
    recompute kernel of the train path with the running statistics as its constants (`kp_conv_first_fwd_bnact`).

    Built from a `knn.Unit` in a 16-bit precision on the device; raises NotImplementedError otherwise, or when no layer of the Unit
    would take the fused kernel at all.  Layers the kernel declines (odd map sizes) run as the plain pair kernel + pass.  The
    scale / shift vectors are a snapshot of the BatchNorm buffers: build a new one after changing parameters or buffers."""

    def __init__(self, unit):
        from .runtime import precision_dtype
        if unit.precision not in ('bf16', 'fp16'):
            raise NotImplementedError('fused inference (matrix-core form): 16-bit storage only')
        plan = unit.plan()
        convs, head = plan[:-1], plan[-1]
        if not convs:
            raise NotImplementedError('fused inference: no 3x3 layer')
        p0 = next(unit.parameters())
        if p0.device.type != 'cuda':
            raise RuntimeError('FusedUnitF: the Unit must live on a HIP device (no CPU fallback)')
        if not any(s['conv'].in_channels % 64 == 0 and s['conv'].out_channels % 64 == 0 for s in convs):
            raise NotImplementedError('fused inference (matrix-core form): no layer with 64-multiple channels')
        self.unit, self.plan = unit, plan
        self.tdt, self.code = precision_dtype(unit.precision)
        self.cin = convs[0]['conv'].in_channels
        self.K = head['conv'].out_channels
        self._packs = {}
        self.consts = []
        with torch.no_grad():
            for s in convs:
                conv, bn = s['conv'], s['bn']
                b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(conv.out_channels, device=p0.device)
                if bn is None:
                    scale, shift, mean, invstd = torch.ones_like(b), b.clone(), None, None
                else:
                    if bn.running_mean is None or bn.running_var is None:
                        raise NotImplementedError('fused inference: BatchNorm without running statistics')
                    invstd = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
                    mean = bn.running_mean.detach().float().clone()
                    scale = bn.weight.detach().float() * invstd
                    shift = bn.bias.detach().float() + (b - mean) * scale
                self.consts.append(dict(scale=scale.contiguous(), shift=shift.contiguous(), mean=mean, invstd=invstd))
        self.fused_layers = 0     # layers of the LAST call that ran on kp_conv_fwd_act / kp_conv_first_fwd_bnact (diagnostics, tests)

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        from .unit_engine import TAILS, _first_cp, _pack_unit, _tail_dims
        if not x.is_cuda or x.dim() != 4 or x.shape[1] != self.cin:
            raise RuntimeError(f'FusedUnitF: expected a HIP tensor [B, {self.cin}, H, W]')
        x = x.float().contiguous()
        plan, tdt, code = self.plan, self.tdt, self.code
        L = _lib.lib()
        st = stream_ptr()
        dev = x.device
        N, Cin, H, W = x.shape
        Cp0 = _first_cp(plan, Cin, self.unit.precision)
        a = torch.empty(N, H + 2, W + 2, Cp0, device=dev, dtype=tdt)
        call('kp_nchw_to_nhwc', ptr(x), ptr(a), N, Cin, Cp0, H, W, 1, code, st)
        with shared_packs(self._packs):
            packs = _pack_unit(plan, Cin, Cp0, False, dev, tdt, code, st)
        C = Cin
        fused = 0
        for idx, stp in enumerate(plan[:-1]):
            conv, bn, slope, tail = stp['conv'], stp['bn'], stp['slope'], stp['tail']
            K = conv.out_channels
            Cphys = Cp0 if idx == 0 else C
            wf = packs[idx][0]
            cst = self.consts[idx]
            nxt = plan[idx + 1]
            out_pad = 1 if nxt['kind'] == 'conv3' else 0
            Ho, Wo = _tail_dims(H, W, tail)
            if (idx == 0 and bn is not None and tail is None and nxt['kind'] == 'conv3' and Cp0 == Cin and 0.0 <= slope <= 1.0
                    and L.kp_conv_first_fused_supported(N, H, W, Cin, K, code) == 1):
                a_next = torch.empty(N, H + 2, W + 2, K, device=dev, dtype=tdt)
                call('kp_conv_first_fwd_bnact', ptr(a), ptr(wf), ptr(conv.bias), ptr(cst['mean']), ptr(cst['invstd']), ptr(bn.weight), ptr(bn.bias),
                     slope, ptr(a_next), N, H, W, Cin, K, code, st)
                fused += 1
            elif 0.0 <= slope <= 1.0 and L.kp_conv_fwd_act_supported(N, H, W, Cphys, K, code) == 1:
                if tail is None:
                    a_next = torch.empty(N, H + 2 * out_pad, W + 2 * out_pad, K, device=dev, dtype=tdt)
                    call('kp_conv_fwd_act', ptr(a), ptr(wf), ptr(cst['scale']), ptr(cst['shift']), slope, ptr(a_next), out_pad, N, H, W, Cphys, K, code, st)
                    if out_pad:
                        call('kp_pad_border', ptr(a_next), N, H, W, K, out_pad, code, st)
                else:   # the activated map, then MaxPool2d(2, 2) / bilinear x2 (+ the next layer's pad) as the pass kernel with identity constants
                    act = torch.empty(N, H, W, K, device=dev, dtype=tdt)
                    call('kp_conv_fwd_act', ptr(a), ptr(wf), ptr(cst['scale']), ptr(cst['shift']), slope, ptr(act), 0, N, H, W, Cphys, K, code, st)
                    a_next = torch.empty(N, Ho + 2 * out_pad, Wo + 2 * out_pad, K, device=dev, dtype=tdt)
                    call('kp_bn_act_fwd', ptr(act), None, None, None, None, 1.0, TAILS[tail], ptr(a_next), out_pad, None, N, H, W, K, code, st)
                fused += 1
            else:   # a shape the fused kernel declines: the plain convolution + the eval-mode pass (what Unit.forward runs)
                y = torch.empty(N, H, W, K, device=dev, dtype=tdt)
                call('kp_conv_fwd', ptr(a), ptr(wf), ptr(conv.bias), ptr(y), None, N, H + 2, W + 2, Cphys, K, 3, code, st)
                a_next = torch.empty(N, Ho + 2 * out_pad, Wo + 2 * out_pad, K, device=dev, dtype=tdt)
                call('kp_bn_act_fwd', ptr(y), ptr(cst['mean']), ptr(cst['invstd']), ptr(bn.weight if bn is not None else None),
                     ptr(bn.bias if bn is not None else None), slope, TAILS[tail], ptr(a_next), out_pad, None, N, H, W, K, code, st)
            a, C, H, W = a_next, K, Ho, Wo
        head = plan[-1]
        hconv = head['conv']
        out = torch.empty(N, self.K, H, W, device=dev, dtype=torch.float32)
        call('kp_head_fwd', ptr(a), ptr(hconv.weight), ptr(hconv.bias), head['slope'], ptr(out), N, H, W, C, self.K, code, st)
        self.fused_layers = fused
        return out

    def keypoints(self, x: torch.Tensor):
        """`spacial_logsoftmax(unit(x))`: (k [B, K, 2], heat [B, K, H, W])."""
        heat = self(x)
        return MF.spacial_logsoftmax(heat), heat


class KeypointExtractor:
    """`k = KeypointExtractor(net)(x)`: keypoints `[B, K, 2]` (y, x in [0, 1]) of `x [B, C, H, W]`.

    net: a `TransporterNet` / `KeyNet` (anything with a `.keypoint` Unit); it is switched to eval mode.
    graph=True captures one HIP graph per (shape, dtype) of `x`; the returned tensor is then a static buffer that the
    next call with the same shape overwrites (clone it to keep it).  `heatmap` of the last call is kept on the object.
    The weights are taken as frozen: their packed bf16 copies are made once and kept (not part of the graph); call
    `refresh()` after changing them.
    """

    def __init__(self, net, graph: bool = True, warmup: int = 3, fused='auto'):
        """fused: 'auto' = the one-kernel Unit (Pong-class, `FusedUnit`) or the BatchNorm-in-the-epilogue chain (matrix-core Units in
        a 16-bit precision, `FusedUnitF`) when the keypoint network has one of those forms, True = require one, False = always the
        layer-by-layer chain (the training kernels in eval mode, in the network's precision)."""
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
            try:     # not a Pong-class Unit: the matrix-core form (model F's class)
                self._fused_unit = FusedUnitF(self.net.keypoint)
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
