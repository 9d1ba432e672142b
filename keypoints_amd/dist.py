"""Pure data parallelism for the train step: one process per GPU, minibatch sharded along B, one
all-reduce of each Unit's flat gradient arena per step (RCCL over xGMI; `backend="nccl"` is RCCL on ROCm).

The reference has no distributed code; semantics follow the usual DDP reading:
per-rank BatchNorm statistics, gradients averaged over ranks, identical parameters on every rank, BatchNorm buffers
of rank 0 are the ones a checkpoint holds.

Each `knn.Unit` owns one contiguous fp32 gradient buffer (unit_engine.GradArena): the decoder's
(23.9 MB for model F) is complete first and is reduced on RCCL's stream while the keypoint / feature
backward passes still run; `finish()` joins before the optimiser step.

Inside a Unit the backward runs from the deep layers to the first one, and the deep 512-channel layers hold most of
the parameters but little of the time: the arena is therefore reduced in two pieces -- the tail (every layer from the
split on, plus the 1x1 head) as soon as those layers are done, overlapping the long high-resolution layers, and the
small head of the arena at the end of the Unit's backward.  Only that last piece of the last Unit is exposed.

Contract: between two `finish()` calls every Unit is reduced ONCE.  A second backward through a Unit whose pieces are
already in flight (gradient accumulation over micro-batches, two losses) would add into memory an all-reduce is
reading, so it raises; run the earlier backwards under `no_sync()` (they only accumulate into the arena) and the last
one outside of it.
"""
from __future__ import annotations

import contextlib
from typing import List

import torch
import torch.distributed as dist

from . import _lib, runtime


class NativeComm:
    """The RCCL communicator behind the C ABI (`kp_comm_*`, include/kp_hip.h): the gradient exchange without
    torch.distributed in the data path.  torch.distributed (any backend) is only the side channel that carries rank 0's
    128-byte unique id to the other ranks; collectives then run on a communication stream of this object, ordered
    against the compute stream with events.  One per process (= per GPU); bound to the current device."""

    _DT = {torch.float32: _lib.KP_F32, torch.bfloat16: _lib.KP_BF16, torch.float16: _lib.KP_F16}

    def __init__(self, group=None):
        import ctypes
        if not _lib.lib().kp_comm_available():
            raise RuntimeError('NativeComm: no RCCL library in this process (librccl.so)')
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        buf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.call('kp_comm_unique_id', buf)
        box = [bytes(buf.raw) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        h = ctypes.c_void_p()
        _lib.call('kp_comm_init', box[0], self.rank, self.world, ctypes.byref(h))
        self._h = h
        self.stream = torch.cuda.Stream()

    class _Work:
        def __init__(self, ev):
            self._ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self._ev)

    def all_reduce(self, t: torch.Tensor, average: bool = True):
        """In-place all-reduce of a contiguous tensor, asynchronous to the caller's stream: everything enqueued there so
        far happens before it; `.wait()` of the returned object orders the caller's stream behind it."""
        self.stream.wait_stream(torch.cuda.current_stream())
        _lib.call('kp_comm_allreduce', self._h, t.data_ptr(), t.numel(), self._DT[t.dtype], int(average), self.stream.cuda_stream)
        ev = torch.cuda.Event(); ev.record(self.stream)
        return NativeComm._Work(ev)

    def broadcast(self, t: torch.Tensor, src: int = 0) -> None:
        st = torch.cuda.current_stream()
        _lib.call('kp_comm_broadcast', self._h, t.data_ptr(), t.numel(), self._DT[t.dtype], src, st.cuda_stream)

    def count(self) -> int:
        """Number of ranks of the RCCL communicator itself (ncclCommCount)."""
        import ctypes
        n = ctypes.c_int(0)
        _lib.call('kp_comm_count', self._h, ctypes.byref(n))
        return int(n.value)

    def close(self) -> None:
        if self._h is not None:
            torch.cuda.synchronize()
            _lib.call('kp_comm_destroy', self._h)
            self._h = None


class GradReducer:
    def __init__(self, root: torch.nn.Module, group=None, broadcast_init: bool = True, tail_fraction: float = 0.6,
                 transport: str = 'fp32', overlap: bool = True, sync_bn: bool = False, comm: str = 'torch', reserve_cus: int = 32,
                 reserve_alone: bool = False):
        """overlap: True -- pieces are all-reduced from the backward hooks, concurrently with the rest of the backward;
        False -- the hooks only note which arenas are final and `finish()` reduces them back to back (the exchange is then
        fully exposed, ~1 ms for model F over xGMI, but no RCCL kernel ever shares the CUs with the persistent MFMA
        kernels, whose tile partition is static: see DESIGN.md section 7).
        reserve_cus: while overlap is True the persistent kernels of this process size their grids to `device CUs - reserve_cus`
        (kp_config(KP_CFG_CUS): a FIXED grid, so tile lists and BatchNorm statistics rows stay a pure function of the
        configuration -- data bit-identical to the full grid, statistics equal up to the order of fp32 additions), which leaves
        that many CUs -- 4 per XCD at the default 32 -- permanently free for RCCL's channel workgroups: a collective that starts in
        the middle of a persistent launch finds CUs of its own instead of queueing workgroups of the NEXT launch behind it.  Cap
        RCCL to as many channels (`NCCL_MAX_NCHANNELS=<reserve_cus>` in the launcher's environment, before the communicator is
        created: bench.py --dp-overlap 1 does).  0 = the whole device for the kernels (round 4's behaviour).  The knob is process-global, so
        the reservation is made only with more than one rank (`reserve_alone=True`: also for a lone rank -- the single-GPU measurement
        of what the reduced grid costs) and the previous value of the knob is restored when overlap is switched off / `detach()`.
        Why 32: measured on one GPU (bench.py --grid-cus, profiles/r05c_grid_cus.txt), model F fp16 step: 256 CUs 33.08 ms, 248
        34.39, 240 34.06, 224 33.66 -- the tile lists of the BASELINE layers divide evenly over 28 workgroups per XCD and not over
        30 or 31, so 224 costs 1.8 % where 240 costs 3 %.
        transport: 'fp32' (exact mean of the fp32 gradients) or 'bf16' (each piece is rounded to bf16 for the wire
        and the reduction: half the xGMI bytes, gradient error 2^-9 relative per element -- SURVEY.md 8e option).
        comm: 'torch' -- the all-reduces go through torch.distributed (ProcessGroupNCCL = RCCL on ROCm, or gloo);
        'native' -- through the C ABI's own RCCL communicator (`NativeComm`, kp_comm_*: SURVEY.md 8b), fp32 / bf16 parameters
        and gradients only; torch.distributed then only carries the communicator's unique id at construction.
        sync_bn: BatchNorm batch statistics (forward) and the two backward sums are all-reduced per layer, i.e. the ranks
        normalise with the statistics of the GLOBAL batch: the DP step then equals the single-process step on the
        concatenated batch (non-default option; one small synchronous all-reduce per layer and direction)."""
        from .models.knn import Unit
        if transport not in ('fp32', 'bf16'):
            raise ValueError("transport must be 'fp32' or 'bf16'")
        self.root = root
        self.group = group
        self.world = dist.get_world_size(group)
        self.transport = transport
        self.reserve_cus = max(0, int(reserve_cus))
        self.reserve_alone = bool(reserve_alone)
        self.grid_cus = None     # CU budget of the persistent kernels while this reducer holds a reservation (None: it holds none)
        self._cus_before = 0     # what KP_CFG_CUS held when the reservation was made (restored when it ends)
        self._overlap = False
        self.overlap = overlap   # (property: reserves / releases the CUs)
        self._ready = []     # overlap=False: arena slices noted by the hooks, reduced in finish()
        self.units: List[Unit] = [m for m in root.modules() if isinstance(m, Unit)]
        self._pending = []   # (work, fp32 arena slice, wire tensor or None)
        if comm not in ('torch', 'native'):
            raise ValueError("comm must be 'torch' or 'native'")
        self.native = NativeComm(group) if comm == 'native' else None
        self._avg = self.native is not None or dist.get_backend(group) == 'nccl'   # RCCL has ncclAvg; gloo sums then scales
        self._sync = True
        self.pieces_issued = 0   # since the last finish() (tests / diagnostics)
        self.collectives_last_finish = 0   # all-reduces the last finish() waited for
        self.timing = None       # list of (start, end) HIP event pairs around finish() while `time_finish(True)` (bench.py)
        if broadcast_init:
            for t in list(root.parameters()) + list(root.buffers()):
                self._bcast(t.data)
        self._split = {}      # id(unit) -> (plan step index, arena offset): the tail from there on is reduced early
        self._early = set()   # units whose tail piece is already in flight this step
        self._done = set()    # units whose whole arena is in flight this step
        for u in self.units:
            u.after_backward = self._on_unit_backward
            u.layer_done = self._on_layer_done
            u.sync_bn = (group, self.world) if sync_bn and self.world > 1 else None
            self._split[id(u)] = self._choose_split(u, tail_fraction)

    @property
    def overlap(self) -> bool:
        return self._overlap

    @overlap.setter
    def overlap(self, on: bool) -> None:
        on = bool(on)
        p0 = next(self.root.parameters(), None)
        on_device = p0 is not None and p0.is_cuda   # (the CPU / gloo tests of the bookkeeping have no library knob to set)
        # KP_CFG_CUS is process-global: the reservation is made only where a collective can actually run beside the kernels (more than
        # one rank -- a lone rank would pay the reduced grid's ~1.8 % for nothing -- or `reserve_alone` for the single-GPU measurement
        # of that cost), and what the knob held before (a budget the caller set, bench.py --grid-cus; another reducer's) is put back
        # when the reservation ends instead of being overwritten with 0
        want = None
        if on and self.reserve_cus > 0 and on_device and (self.world > 1 or self.reserve_alone):
            total = torch.cuda.get_device_properties(p0.device).multi_processor_count
            # a multiple of the 8 XCDs (the persistent kernels round their grids down to one anyway), never below one CU per XCD
            want = max(8, (total - self.reserve_cus) // 8 * 8)
        if on_device:
            if want is not None:
                if self.grid_cus is None:
                    self._cus_before = int(_lib.lib().kp_config_get(_lib.KP_CFG_CUS))
                _lib.call('kp_config', _lib.KP_CFG_CUS, want)
            elif self.grid_cus is not None:
                _lib.call('kp_config', _lib.KP_CFG_CUS, max(0, self._cus_before))
        self.grid_cus = want
        self._overlap = on

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
            off = arena.views[id(plan[idx]['conv'].weight)].storage_offset() - arena.flat.storage_offset()   # (the arena may be a slice)
            if total - off >= tail_fraction * total:
                best = (idx, off)
        return best

    @contextlib.contextmanager
    def no_sync(self):
        """Backwards inside only accumulate into the gradient arenas (micro-batch accumulation); nothing is sent."""
        prev, self._sync = self._sync, False
        try:
            yield self
        finally:
            self._sync = prev

    def _reduce(self, t) -> None:
        if not self.overlap:
            self._ready.append(t)
            self.pieces_issued += 1
            return
        self._launch(t)
        self.pieces_issued += 1

    def _bcast(self, t, src: int = 0) -> None:
        if self.native is not None and t.dtype in NativeComm._DT and t.is_contiguous():
            self.native.broadcast(t, src)
        else:   # (num_batches_tracked is int64: RCCL types of the C ABI are the float ones)
            dist.broadcast(t, src=src, group=self.group)

    def _launch(self, t) -> None:
        wire = t.to(torch.bfloat16) if self.transport == 'bf16' else None
        if self.native is not None:
            work = self.native.all_reduce(wire if wire is not None else t, average=True)
        else:
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            work = dist.all_reduce(wire if wire is not None else t, op=op, group=self.group, async_op=True)
        self._pending.append((work, t, wire))

    def _check_not_in_flight(self, unit) -> None:
        if id(unit) in self._done or id(unit) in self._early:
            raise RuntimeError(
                'GradReducer: a second backward reached a Unit whose gradient arena is already being all-reduced '
                '(no finish() in between).  Run all but the last backward of a step under reducer.no_sync().')

    def _on_layer_done(self, unit, idx) -> None:
        """Fires after each 3x3 layer of a Unit's backward (deepest first): at the split, reduce the arena's tail."""
        if not self._sync:
            return
        if id(unit) in self._done:
            self._check_not_in_flight(unit)
        sp = self._split.get(id(unit)) if self.overlap else None   # overlap=False: one all-reduce per whole arena
        if sp is not None and idx == sp[0]:
            self._check_not_in_flight(unit)
            self._early.add(id(unit))
            self._reduce(unit.grad_arena().flat[sp[1]:])

    def _on_unit_backward(self, unit) -> None:
        """Fires at the end of a Unit's backward: its arena is final -> reduce what is not in flight yet."""
        if not self._sync:
            return
        if id(unit) in self._done:
            self._check_not_in_flight(unit)
        flat = unit.grad_arena().flat
        if id(unit) in self._early:
            self._early.discard(id(unit))
            self._reduce(flat[:self._split[id(unit)][1]])
        else:
            self._reduce(flat)
        self._done.add(id(unit))

    @staticmethod
    def _coalesce(ts):
        """overlap=False: the arenas noted during the backward, merged where they are neighbours in memory -- FusedAdam.bind keeps
        every Unit's gradients in ONE buffer (slices a few zero floats apart), so the whole network goes out as one collective
        instead of one per Unit (fewer, larger all-reduces: what a ring over xGMI wants).  Anything else passes through unchanged."""
        flat = [t for t in ts if t.dim() == 1 and t.is_contiguous()]
        rest = [t for t in ts if not (t.dim() == 1 and t.is_contiguous())]
        groups = {}
        for t in flat:
            groups.setdefault((t.untyped_storage().data_ptr(), t.dtype, t.device), []).append(t)
        out = []
        for g in groups.values():
            g.sort(key=lambda t: t.storage_offset())
            run = [g[0]]
            for t in g[1:]:
                prev = run[-1]
                gap = t.storage_offset() - (prev.storage_offset() + prev.numel())
                if 0 <= gap < 64:
                    run.append(t)
                else:
                    out.append(run); run = [t]
            out.append(run)
        merged = []
        for run in out:
            if len(run) == 1:
                merged.append(run[0])
            else:
                first, last = run[0], run[-1]
                n = last.storage_offset() + last.numel() - first.storage_offset()
                merged.append(torch.empty(0, dtype=first.dtype, device=first.device).set_(
                    first.untyped_storage(), first.storage_offset(), (n,), (1,)))
        return merged + rest

    def ranks_seen(self) -> dict:
        """How many ranks the gradient exchange really spans, asked of the transport itself rather than of the launcher's
        environment: a SUM all-reduce of a one on the device through the very communicator / process group the gradients
        use (every participating rank contributes 1), plus ncclCommCount for the native communicator.  bench.py prints this
        as `dp.nccl_ranks` so that a multi-GPU record proves the collective saw N ranks."""
        dev = next(self.root.parameters()).device
        one = torch.ones(1, device=dev, dtype=torch.float32)
        if self.native is not None:
            self.native.all_reduce(one, average=False).wait()
            out = dict(allreduce_of_ones=int(round(float(one.item()))), comm_count=self.native.count(), transport='kp_comm (RCCL via the C ABI)')
        else:
            dist.all_reduce(one, op=dist.ReduceOp.SUM, group=self.group)
            out = dict(allreduce_of_ones=int(round(float(one.item()))), comm_count=dist.get_world_size(self.group),
                       transport=f'torch.distributed {dist.get_backend(self.group)}')
        return out

    def time_finish(self, on: bool) -> None:
        """Record a HIP event pair around every finish() (the EXPOSED part of the exchange: what the step waits for between
        backward and the optimiser)."""
        self.timing = [] if on else None

    def exposed_ms(self) -> List[float]:
        """Milliseconds between the two events of each timed finish() (synchronises)."""
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in (self.timing or [])]

    def finish(self) -> None:
        """Join every outstanding gradient all-reduce (call between backward() and optimiser.step()).  Every piece is
        a disjoint arena slice, reduced and scaled exactly once."""
        runtime.join_side_streams()
        if self.timing is not None:
            t0 = torch.cuda.Event(enable_timing=True); t0.record()
        if self._early:
            raise RuntimeError('GradReducer.finish(): a Unit sent its arena tail but never finished its backward')
        for t in self._coalesce(self._ready):
            self._launch(t)
        self._ready.clear()
        for work, flat, wire in self._pending:
            work.wait()
            if wire is not None:
                flat.copy_(wire)
            if not self._avg:
                flat.div_(self.world)
        self.collectives_last_finish = len(self._pending)
        self._pending.clear()
        self._done.clear()
        self.pieces_issued = 0
        if self.timing is not None:
            t1 = torch.cuda.Event(enable_timing=True); t1.record()
            self.timing.append((t0, t1))

    def broadcast_buffers(self, src: int = 0) -> None:
        """BatchNorm running statistics are per-rank (each rank normalises its own shard); before a checkpoint every
        rank takes rank `src`'s buffers, so that what is saved -- and what every rank continues from -- is one state."""
        for b in self.root.buffers():
            self._bcast(b.data, src)

    def save(self, directory: str) -> None:
        """Checkpoint under DP: buffers from rank 0 to everyone, then rank 0 writes `root.save(directory)`
        (the reference's file layout, transporter.py:66-70 / keynet.py:33-37)."""
        self.broadcast_buffers(0)
        if dist.get_rank(self.group) == 0:
            self.root.save(directory)
        dist.barrier(group=self.group)

    def detach(self) -> None:
        self.overlap = False   # (gives the reserved CUs back to the persistent kernels)
        if self.native is not None:
            self.native.close()
            self.native = None
        for u in self.units:
            u.after_backward = None
            u.layer_done = None
            u.sync_bn = None
