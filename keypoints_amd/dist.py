"""Pure data parallelism for the train step: one process per GPU, minibatch sharded along B, one
all-reduce of each Unit's flat gradient arena per step (RCCL over xGMI; `backend="nccl"` is RCCL on ROCm).

The reference has no distributed code; semantics follow the usual DDP reading:
per-rank BatchNorm statistics, gradients averaged over ranks, identical parameters on every rank.

Each `knn.Unit` owns one contiguous fp32 gradient buffer (unit_engine.GradArena): the decoder's
(23.9 MB for model F) is complete first and is reduced on RCCL's stream while the keypoint / feature
backward passes still run; `finish()` joins before the optimiser step.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, root: torch.nn.Module, group=None, broadcast_init: bool = True):
        from .models.knn import Unit
        self.group = group
        self.world = dist.get_world_size(group)
        self.units: List[Unit] = [m for m in root.modules() if isinstance(m, Unit)]
        self._pending = []
        self._avg = dist.get_backend(group) == 'nccl'   # RCCL has ncclAvg; gloo sums then scales
        if broadcast_init:
            for t in list(root.parameters()) + list(root.buffers()):
                dist.broadcast(t.data, src=0, group=group)
        for u in self.units:
            u.after_backward = self._on_unit_backward

    def _on_unit_backward(self, unit) -> None:
        """Fires at the end of a Unit's backward: its arena is final -> launch its all-reduce now."""
        flat = unit.grad_arena().flat
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        work = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        self._pending.append((work, flat))

    def finish(self) -> None:
        """Join every outstanding gradient all-reduce (call between backward() and optimiser.step())."""
        for work, flat in self._pending:
            work.wait()
            if not self._avg:
                flat.div_(self.world)
        self._pending.clear()

    def detach(self) -> None:
        for u in self.units:
            u.after_backward = None
