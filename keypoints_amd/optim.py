"""Loss and optimiser of the train step (transporter.py:47,56-60,79-89) on HIP kernels.

`l2_reconstruction_loss` is the reference's criterion; `FusedAdam` replaces
``torch.optim.Adam(params, lr=1e-4)`` with one kernel launch per flat parameter arena.
"""
from __future__ import annotations

from typing import Iterable, List

import torch

from . import runtime
from ._lib import call, ptr
from .runtime import f32c, require_device, stream_ptr


_LOSS_WS = {}


def _loss_ws(device):
    """The two-double workspace of kp_l2_loss_mean for torch's current stream on `device` (zero at creation; every call leaves it
    zero, so calls on one stream share it; calls on different streams may overlap and get their own)."""
    key = (device.index, torch.cuda.current_stream(device).stream_id)
    ws = _LOSS_WS.get(key)
    if ws is None:
        ws = torch.zeros(2, device=device, dtype=torch.float64)
        # created while a HIP graph is being captured, the zero-fill is only RECORDED: such a workspace belongs to that graph (its
        # replays zero it first) and is not handed to eager calls on the same stream id, which would find it uninitialised
        if not torch.cuda.is_current_stream_capturing():
            _LOSS_WS[key] = ws
    return ws


class _L2Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, mask):
        require_device(x, target, mask)
        x, target = f32c(x), f32c(target)
        if mask is not None:
            mask = f32c(mask).expand_as(x).contiguous()
        n = x.numel()
        grad = torch.empty_like(x)
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        # (the mean -- fp64 sum / n, rounded once to fp32 -- is written by the kernel's last workgroup: no zero-fill, no division
        #  launch between the forward and the backward)
        try:
            call('kp_l2_loss_mean', ptr(x), ptr(target), ptr(mask), ptr(_loss_ws(x.device)), ptr(loss), ptr(grad), n, stream_ptr())
        except Exception:
            # a launch that did not complete may leave the sum / arrival counter non-zero: never reuse that workspace
            _LOSS_WS.pop((x.device.index, torch.cuda.current_stream(x.device).stream_id), None)
            raise
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        gs = f32c(g).reshape(1)                       # upstream gradient of the scalar loss (device tensor)
        out = torch.empty_like(grad)
        call('kp_scale', ptr(grad), ptr(gs), ptr(out), grad.numel(), stream_ptr())
        return out, None, None


def l2_reconstruction_loss(x, x_, loss_mask=None):
    """mean((x - x_)^2 [* loss_mask]); gradient flows to `x` only (x_ is data)."""
    from .models.functional import unlazy
    x, x_, loss_mask = unlazy(x), unlazy(x_), unlazy(loss_mask)
    return _L2Loss.apply(x, x_.detach(), loss_mask)


class FusedAdam:
    """Adam (betas .9/.999, eps 1e-8, no weight decay -- torch.optim.Adam defaults) over flat arenas.

    Parameters are re-homed into one flat fp32 buffer per group of parameters that share a
    GradArena (one per `knn.Unit`), so a step is one `kp_adam_step` launch per Unit and the
    gradient buffer is the data-parallel all-reduce bucket.  Loose parameters (not owned by a
    Unit) get one launch each.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-4, betas=(0.9, 0.999), eps=1e-8, modules=None):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps)]
        self._units = []
        self._whole = []   # (params, grads, exp_avg, exp_avg_sq) of all Units of a device as one buffer each (bind)
        self._loose = []
        self._state = {}
        self.scaler = None   # keypoints_amd.amp.LossScaler for the fp16 mode (amp.initialize attaches it)
        if modules is not None:
            self.bind(modules)

    def bind(self, root: torch.nn.Module) -> 'FusedAdam':
        """Flatten the parameters of every `Unit` under `root` (call once, after `.to(device)` and before the first step: binding
        again re-homes the parameters and starts the Adam moments from zero)."""
        from .models.knn import Unit
        from .unit_engine import GradArena
        self._units.clear(); self._whole.clear()   # (a second bind re-homes everything again: no Unit is stepped twice)
        owned = set()
        wanted = {id(p) for p in self.params}
        todo = []
        for m in root.modules():
            if isinstance(m, Unit):
                ps = [p for p in m.parameters()]
                if not ps or not all(id(p) in wanted for p in ps) or any(id(p) in owned for p in ps):
                    continue
                require_device(*ps)
                todo.append((m, ps))
                owned.update(id(p) for p in ps)
        # ONE parameter / gradient / moment buffer per device for all Units (a Unit's arena is a 256-byte-aligned slice of it):
        # the step and zero_grad are then one launch each for the whole network, whatever the number of Units
        by_dev = {}
        for m, ps in todo:
            by_dev.setdefault(ps[0].device, []).append((m, ps))
        for dev, group in by_dev.items():
            sizes = [sum(p.numel() for p in ps) for _, ps in group]
            starts, total = [], 0
            for n in sizes:
                starts.append(total)
                total += (n + 63) // 64 * 64
            w_all = torch.zeros(total, device=dev, dtype=torch.float32)
            g_all, m_all, v_all = torch.zeros_like(w_all), torch.zeros_like(w_all), torch.zeros_like(w_all)
            self._whole.append((w_all, g_all, m_all, v_all))
            for (m, ps), s0, n in zip(group, starts, sizes):
                flat = w_all[s0:s0 + n]
                off = 0
                for p in ps:
                    view = flat[off:off + p.numel()].view_as(p)
                    view.copy_(p.data)
                    p.data = view
                    off += p.numel()
                m._arena = GradArena(ps, flat=g_all[s0:s0 + n])   # grad views in the same ordering
                self._units.append((m, flat, m_all[s0:s0 + n], v_all[s0:s0 + n]))
        self._loose = [p for p in self.params if id(p) not in owned]
        return self

    def _whole_ok(self):
        """True while every Unit still accumulates into its slice of the shared gradient buffer (a Unit moved to another device,
        or given a fresh arena by hand, takes the per-Unit path again)."""
        return bool(self._whole) and all(m._arena is not None and m._arena.flat.data_ptr() == g.data_ptr()
                                         for (m, *_), g in zip(self._units, self._grad_slices()))

    def _grad_slices(self):
        out = []
        for w_all, g_all, *_ in self._whole:
            for m, flat, *_ in self._units:
                if flat.device == w_all.device:
                    off = (flat.data_ptr() - w_all.data_ptr()) // 4
                    out.append(g_all[off:off + flat.numel()])
        return out

    def zero_grad(self, set_to_none: bool = False):
        whole = self._whole_ok()
        if whole:
            for _, g_all, *_ in self._whole:
                g_all.zero_()
        for m, *_ in self._units:
            arena = m.grad_arena()
            if not whole:
                arena.flat.zero_()
            for p in m.parameters():
                p.grad = arena.views[id(p)]
        for p in self._loose:
            if p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        runtime.join_side_streams()   # (a backward joins them itself when it ends; this is for hand-driven Units)
        if not self._units and not self._loose:
            self._loose = list(self.params)
        self.step_count += 1
        t = self.step_count
        # hyper-parameters are read from param_groups[0] every step, like torch.optim.Adam: LR schedulers and manual
        # edits of `opt.param_groups[0]['lr']` take effect (one group: the arenas are per Unit, not per group)
        if len(self.param_groups) != 1:
            raise RuntimeError(f'FusedAdam: {len(self.param_groups)} param_groups -- one is supported (the flat arenas are per '
                               'Unit, not per group); parameters of an added group would silently never be updated')
        grp = self.param_groups[0]
        self.lr, self.betas, self.eps = grp['lr'], grp['betas'], grp['eps']
        b1, b2 = self.betas
        bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
        st = stream_ptr()
        if self._whole_ok():   # (the padding between the Units' slices is zero in all four buffers and stays zero)
            work = list(self._whole)
        else:
            work = [(flat, m.grad_arena().flat, em, ev) for m, flat, em, ev in self._units]
        for p in self._loose:
            if p.grad is None:
                continue
            require_device(p)
            stt = self._state.setdefault(id(p), (torch.zeros_like(p.data), torch.zeros_like(p.data)))
            work.append((p.data, f32c(p.grad), stt[0], stt[1]))
        if self.scaler is None:
            for w, g, em, ev in work:
                call('kp_adam_step', ptr(w), ptr(g), ptr(em), ptr(ev), w.numel(), self.lr, b1, b2, self.eps, bc1, bc2, st)
            return
        # fp16 mode: gradients carry the loss scale.  Flag inf / nan anywhere, then every arena either takes the
        # un-scaled Adam update or (overflow) is left alone, then the scale adapts -- all on the device, no host read.
        sc = self.scaler
        for w, g, em, ev in work:
            call('kp_amp_check', ptr(g), g.numel(), ptr(sc.state), st)
        call('kp_amp_prepare', ptr(sc.state), b1, b2, st)
        for w, g, em, ev in work:
            call('kp_adam_step_scaled', ptr(w), ptr(g), ptr(em), ptr(ev), w.numel(), self.lr, b1, b2, self.eps,
                 ptr(sc.state), st)
        call('kp_amp_update', ptr(sc.state), sc.growth_factor, sc.backoff_factor, sc.growth_interval, sc.max_scale, st)
