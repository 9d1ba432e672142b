"""Pure data parallelism for the train step: one process per GPU, minibatch sharded along B, one
all-reduce of each Unit's flat gradient arena per step (RCCL over xGMI; `backend="nccl"` is RCCL on ROCm).

The reference has no distributed code; semantics follow the usual DDP reading:
per-rank BatchNorm statistics, gradients averaged over ranks, identical parameters on every rank.

Each `knn.Unit` owns one contiguous fp32 gradient buffer (unit_engine.GradArena): the decoder's
(23.9 MB for model F) is complete first and is reduced on RCCL's stream while the keypoint / feature
backward passes still run; `finish()` joins before the optimiser step.

Inside a Unit the backward runs from the deep layers to the first one, and the deep 512-channel layers hold most of
the parameters but little of the time: the arena is therefore reduced in two pieces -- the tail (every layer from the
split on, plus the 1x1 head) as soon as those layers are done, overlapping the long high-resolution layers, and the
small head of the arena at the end of the Unit's backward.  Only that last piece of the last Unit is exposed.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, root: torch.nn.Module, group=None, broadcast_init: bool = True, tail_fraction: float = 0.6):
        from .models.knn import Unit
        self.group = group
        self.world = dist.get_world_size(group)
        self.units: List[Unit] = [m for m in root.modules() if isinstance(m, Unit)]
        self._pending = []
        self._avg = dist.get_backend(group) == 'nccl'   # RCCL has ncclAvg; gloo sums then scales
        if broadcast_init:
            for t in list(root.parameters()) + list(root.buffers()):
                dist.broadcast(t.data, src=0, group=group)
        self._split = {}     # id(unit) -> (plan step index, arena offset): the tail from there on is reduced early
        self._early = set()  # units whose tail piece is already in flight this step
        for u in self.units:
            u.after_backward = self._on_unit_backward
            u.layer_done = self._on_layer_done
            self._split[id(u)] = self._choose_split(u, tail_fraction)

    @staticmethod
    def _choose_split(unit, tail_fraction):
        """The deepest plan step whose tail (the layers from it on + the head) still holds >= tail_fraction of the
        parameters -- the earliest moment in the backward at which that much of the arena is final; at least one layer
        stays in front of it.  None if no such step exists (the whole arena is then reduced at the end)."""
        arena = unit.grad_arena()
        plan = unit.plan()
        total = arena.flat.numel()
        best = None
        for idx in range(1, len(plan) - 1):
            off = arena.views[id(plan[idx]['conv'].weight)].storage_offset()
            if total - off >= tail_fraction * total:
                best = (idx, off)
        return best

    def _reduce(self, t) -> None:
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        work = dist.all_reduce(t, op=op, group=self.group, async_op=True)
        self._pending.append((work, t))

    def _on_layer_done(self, unit, idx) -> None:
        """Fires after each 3x3 layer of a Unit's backward (deepest first): at the split, reduce the arena's tail."""
        sp = self._split.get(id(unit))
        if sp is not None and idx == sp[0] and id(unit) not in self._early:
            self._early.add(id(unit))
            self._reduce(unit.grad_arena().flat[sp[1]:])

    def _on_unit_backward(self, unit) -> None:
        """Fires at the end of a Unit's backward: its arena is final -> reduce what is not in flight yet."""
        flat = unit.grad_arena().flat
        if id(unit) in self._early:
            self._early.discard(id(unit))
            self._reduce(flat[:self._split[id(unit)][1]])
        else:
            self._reduce(flat)

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
            u.layer_done = None
