"""Inference-only keypoint extractor.

The downstream consumers of a trained Transporter run only `net.keypoint(x)` + `spacial_logsoftmax` with BatchNorm in
eval mode, one small batch at a time (atari_demo.py:35-36, tests/test_gl_hack.py:454-456 of the reference).  At that
size the ~60 kernels of a Pong-class keypoint network are launch-bound, so the chain is captured once per input shape
as a HIP graph (`torch.cuda.CUDAGraph` is hipGraph on ROCm; the C-ABI calls enqueue on the capturing stream like any
other) and replayed: one host call per frame instead of one per kernel.
"""
from __future__ import annotations

import torch

from .models import functional as MF
from .unit_engine import shared_packs


class KeypointExtractor:
    """`k = KeypointExtractor(net)(x)`: keypoints `[B, K, 2]` (y, x in [0, 1]) of `x [B, C, H, W]`.

    net: a `TransporterNet` / `KeyNet` (anything with a `.keypoint` Unit); it is switched to eval mode.
    graph=True captures one HIP graph per (shape, dtype) of `x`; the returned tensor is then a static buffer that the
    next call with the same shape overwrites (clone it to keep it).  `heatmap` of the last call is kept on the object.
    The weights are taken as frozen: their packed bf16 copies are made once and kept (not part of the graph); call
    `refresh()` after changing them.
    """

    def __init__(self, net, graph: bool = True, warmup: int = 3):
        self.net = net.eval()
        self.graph = graph
        self.warmup = warmup
        self._captured = {}
        self._packs = {}
        self.heatmap = None

    def refresh(self):
        """Forget the packed weights and the captured graphs (the network's parameters changed)."""
        self._captured.clear()
        self._packs.clear()

    @torch.no_grad()
    def _run(self, x):
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
