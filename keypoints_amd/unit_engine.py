"""Runs one `knn.Unit` (pad+conv3x3+BN+act stacks with max-pool / bilinear-up tails and a 1x1 head)
forward and backward entirely through the HIP kernels of libkp_hip.so.

Reference semantics: keypoints/models/knn.py:110-130 (Unit), vgg.py:16-39 (make_layers).
Data layout in HBM: activations are NHWC in the compute dtype with the
replicate border physically present, so every convolution -- forward, data-gradient and
weight-gradient -- is a plain "valid" correlation with no boundary logic:

    a_{l}  [N][H+2][W+2][C]   replicate-padded activation feeding conv l   (written by bn_act_fwd)
    y_{l}  [N][H][W][K]       conv output before BatchNorm                  (written by conv_fwd)
    dy_{l} [N][H+4][W+4][K]   dL/dy_l with a zero border of 2               (written by bn_act_bwd_apply)
    gx_{l} [N][H+2][W+2][C]   dL/da_l over the padded domain                (written by conv_fwd(dy, flipped w))

The adjoint of the replicate pad (folding gx's border onto the edge pixels) happens inside the
consumer of gx (bn_act_bwd_* of layer l-1).
"""
from __future__ import annotations

import contextlib
import contextvars
from typing import List

import torch
import torch.nn as nn

from . import _lib
from ._lib import call, ptr, try_call
from . import runtime
from .runtime import f32c, precision_dtype, require_device, stream_ptr

TAILS = {None: _lib.TAIL_NONE, 'M': _lib.TAIL_MAXPOOL2, 'U': _lib.TAIL_UPSAMPLE2}


# Packed (bf16, tap-major) copies of the conv weights shared between the Unit calls of ONE network forward: the
# Transporter runs every encoder twice (source / target, transporter.py:36-39) on the same weights, which cannot
# change inside a forward.  None outside of `shared_packs()`.  Context-local (contextvars): two networks driven from two
# threads / tasks of one process each see their own store.
_PACKS_VAR = contextvars.ContextVar('keypoints_amd_packs', default=None)


def _packs():
    return _PACKS_VAR.get()


class shared_packs:
    """Context manager: Unit forwards inside it pack each conv weight once (forward and data-gradient layouts).
    `store`: a dict that outlives the context (frozen weights: the inference extractor keeps its packs across frames)."""

    def __init__(self, store=None):
        self._store = store

    def __enter__(self):
        outer = _PACKS_VAR.get()
        self._token = _PACKS_VAR.set(self._store if self._store is not None else (outer if outer is not None else {}))
        return self

    def __exit__(self, *exc):
        _PACKS_VAR.reset(self._token)
        return False


def _slope_of(act: nn.Module) -> float:
    if isinstance(act, nn.LeakyReLU):
        return float(act.negative_slope)
    if isinstance(act, nn.ReLU):
        return 0.0
    raise NotImplementedError(f'activation {type(act).__name__} has no HIP kernel (ReLU / LeakyReLU only)')


def _first_cp(plan, Cin, precision):
    """Physical input channels of a Unit's first 3x3 layer: an odd input width (KeyNet decoder: z + K = 74 channels, keynet.py:55)
    is zero-padded to a multiple of 64 so that the 16-bit MFMA kernels take the layer; its weights are zero-padded to match."""
    first = plan[0]
    if (first['kind'] == 'conv3' and precision in ('bf16', 'fp16') and Cin > 4 and Cin % 64 != 0 and Cin not in (16, 32)
            and first['conv'].out_channels % 64 == 0):   # (16 / 32 channels have their own kernels)
        return (Cin + 63) // 64 * 64
    return Cin


def _collect_packs(plan, Cin, Cp0, record, dev, tdt, code):
    """Pack buffers of every 3x3 weight of a Unit (from the shared store where they exist) + the list of layers still to pack."""
    out, todo = [], []
    C = Cin
    packs = _packs()
    for idx, stp in enumerate(plan[:-1]):
        conv = stp['conv']
        K, Cl = conv.out_channels, C
        Cp = Cp0 if idx == 0 else C
        pkey = (id(conv.weight), Cp, code)
        ent = packs.get(pkey) if packs is not None else None
        if ent is None or (record and ent[1] is None):
            wf = torch.empty(K, 3, 3, Cp, device=dev, dtype=tdt)
            wd = torch.empty(Cp, 3, 3, K, device=dev, dtype=tdt) if (record or packs is not None) else None
            ent = (wf, wd)
            todo.append((conv.weight, wf, wd, K, Cl, Cp))
            if packs is not None:
                packs[pkey] = ent
        out.append(ent)
        C = K
    return out, todo


def _launch_packs(todo, code, st):
    import ctypes
    for i in range(0, len(todo), 24):
        part = todo[i:i + 24]
        n = len(part)
        vp, ia = ctypes.c_void_p * n, ctypes.c_int * n
        call('kp_pack_weights_batch', n, vp(*[ptr(t[0]) for t in part]), vp(*[ptr(t[1]) for t in part]),
             vp(*[ptr(t[2]) for t in part]), ia(*[t[3] for t in part]), ia(*[t[4] for t in part]), ia(*[t[5] for t in part]),
             code, st)


def _pack_unit(plan, Cin, Cp0, record, dev, tdt, code, st):
    """Packed copies (compute type, tap-major) of every 3x3 weight of a Unit: `[K][3][3][C]` for the forward and, when a
    backward may follow (or the packs are shared across calls), `[C][3][3][K]` flipped for the data gradient.  Layers whose
    packs are not in the shared store yet are packed by ONE kp_pack_weights_batch launch."""
    out, todo = _collect_packs(plan, Cin, Cp0, record, dev, tdt, code)
    _launch_packs(todo, code, st)
    return out


# Two scheduling choices with an A/B switch (plain module attributes; keypoints_amd._probe sets them for tools/probes -- nothing here
# reads the environment):
PREPACK = True        # the weight packs of all Units of a network in one launch at the start of its forward (`prepack` below)
SMALL_INLINE = True   # the weight gradients of 16 / 32-channel layers on the backward's own stream instead of the 'wgrad' side stream


def prepack(units_cin):
    """Inside `shared_packs()`: the packs of SEVERAL Units in one launch at the start of a network forward, so that a later Unit
    (the decoder behind the bottleneck) does not find its pack launch on the critical path in the middle of the step.
    `units_cin`: [(unit, input channels)].  Units of one storage type share the launch; anything this cannot plan is left to
    the Unit's own forward."""
    if _packs() is None or not PREPACK:
        return
    groups = {}
    for unit, cin in units_cin:
        try:
            plan = unit.plan()
        except NotImplementedError:
            continue
        if not any(True for _ in unit.parameters()):
            continue
        p0 = next(unit.parameters())
        if not p0.is_cuda:
            continue
        tdt, code = precision_dtype(unit.precision)
        _, todo = _collect_packs(plan, cin, _first_cp(plan, cin, unit.precision), True, p0.device, tdt, code)
        groups.setdefault(code, []).extend(todo)
    if not groups:      # (a CPU network: nothing here; the Unit's own forward raises the loud error)
        return
    st = stream_ptr()
    for code, todo in groups.items():
        _launch_packs(todo, code, st)


def build_plan(blocks: List[nn.Sequential]) -> List[dict]:
    """Group the flat module list of in_block + core + out_block into kernel steps."""
    mods = [m for b in blocks for m in b]
    steps, i = [], 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.ReplicationPad2d):
            conv = mods[i + 1] if i + 1 < len(mods) else None
            if not (isinstance(conv, nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
                    and conv.padding == (0, 0) and tuple(m.padding) == (1, 1, 1, 1)):
                raise NotImplementedError('expected ReplicationPad2d(1) + Conv2d(k=3, s=1)')
            i += 2
            bn = None
            if i < len(mods) and isinstance(mods[i], nn.BatchNorm2d):
                bn = mods[i]; i += 1
            slope = _slope_of(mods[i]); i += 1
            tail = None
            if i < len(mods) and isinstance(mods[i], nn.MaxPool2d):
                mp = mods[i]
                if mp.kernel_size not in (2, (2, 2)) or mp.stride not in (2, (2, 2)):
                    raise NotImplementedError('only MaxPool2d(2, 2) has a HIP kernel')
                tail = 'M'; i += 1
            elif i < len(mods) and isinstance(mods[i], (nn.UpsamplingBilinear2d, nn.Upsample)):
                up = mods[i]
                sf = up.scale_factor
                sf = tuple(float(v) for v in sf) if isinstance(sf, (tuple, list)) else (float(sf),) * 2 if sf else None
                if up.mode != 'bilinear' or not up.align_corners or sf != (2.0, 2.0) or up.size is not None:
                    raise NotImplementedError('only bilinear x2 upsampling with align_corners=True '
                                              '(nn.UpsamplingBilinear2d(scale_factor=2), vgg.py:26) has a HIP kernel')
                tail = 'U'; i += 1
            steps.append(dict(kind='conv3', conv=conv, bn=bn, slope=slope, tail=tail))
        elif isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1) and i + 2 == len(mods):
            steps.append(dict(kind='head', conv=m, slope=_slope_of(mods[i + 1])))
            i += 2
        else:
            raise NotImplementedError(f'no HIP kernel sequence for {type(m).__name__} at position {i} of the Unit')
    if not steps or steps[-1]['kind'] != 'head' or any(s['kind'] == 'head' for s in steps[:-1]):
        raise NotImplementedError('a Unit must end in Conv2d(1x1) + activation (knn.py:123-125)')
    return steps


class GradArena:
    """One flat fp32 buffer holding every parameter gradient of a Unit (the DP all-reduce bucket).

    `views[p]` aliases the arena; backward accumulates into the views and installs them as `p.grad`.
    """

    def __init__(self, params: List[nn.Parameter], flat: torch.Tensor = None):
        """flat: a zeroed fp32 buffer of the parameters' total size to use instead of an own one (FusedAdam.bind hands every
        Unit a slice of ONE buffer, so that the optimizer step and zero_grad are one launch each for the whole network)."""
        dev = params[0].device
        n = sum(p.numel() for p in params)
        if flat is not None and (flat.numel() != n or flat.dtype != torch.float32 or flat.device != dev or not flat.is_contiguous()):
            raise ValueError('GradArena: the external buffer does not match the parameters')
        self.flat = flat if flat is not None else torch.zeros(n, device=dev, dtype=torch.float32)
        self.views, off = {}, 0
        for p in params:
            self.views[id(p)] = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def attach(self, params: List[nn.Parameter]) -> None:
        stale = [p for p in params if p.grad is not self.views[id(p)]]
        if len(stale) == len(params):
            self.flat.zero_()
        else:
            for p in stale:
                self.views[id(p)].zero_()
        for p in stale:
            p.grad = self.views[id(p)]


def _allreduce_rows(rows: torch.Tensor, group) -> torch.Tensor:
    """SyncBN (non-default): per-channel partial sums [R][2][K] (fp32) of this rank -> the sum over all ranks as
    TWO fp32 rows (high part, remainder) of the fp64 total, so that the fp64 finalize kernels lose nothing to the fp32
    transport format.  Tiny tensors (2 K doubles per layer); the all-reduce is synchronous on the launch stream."""
    import torch.distributed as dist
    tot = rows.double().sum(0)
    dist.all_reduce(tot, group=group)
    hi = tot.float()
    lo = (tot - hi.double()).float()
    return torch.stack((hi, lo)).contiguous()


def _tail_dims(H, W, tail):
    if tail == 'M':
        return H // 2, W // 2
    if tail == 'U':
        return 2 * H, 2 * W
    return H, W


def head_dims(unit, H, W):
    """(Hh, Wh, Ch, Kout): map size and channels at a Unit's 1x1 head for an H x W input (the pools / upsamplings of its plan)."""
    plan = unit.plan()
    C = None
    for stp in plan[:-1]:
        H, W = _tail_dims(H, W, stp['tail'])
        C = stp['conv'].out_channels
    hconv = plan[-1]['conv']
    return H, W, (C if C is not None else hconv.in_channels), hconv.out_channels


def pair_ok(unit, N, Cin, H, W, out_t=False):
    """True if `UnitFunction._forward_pair` can run source + target frame of N images each through this Unit as one batch: 16-bit
    storage, train-mode BatchNorm with a momentum in every 3x3 layer, no pool / upsample, every layer and the head on a kernel that
    has the pair form AND the fused BatchNorm-input forms the backward of the target half uses (the Pong-class nets of vgg.py:53,66)."""
    if not (runtime.pair_calls() and runtime.fuse_bn_input() and unit.precision in ('bf16', 'fp16') and unit.sync_bn is None):
        return False
    try:
        plan = unit.plan()
    except NotImplementedError:
        return False
    code = precision_dtype(unit.precision)[1]
    L = _lib.lib()
    if len(plan) < 2 or plan[0]['kind'] != 'conv3' or _first_cp(plan, Cin, unit.precision) != Cin:
        return False
    C = Cin
    for idx, stp in enumerate(plan[:-1]):
        bn, K = stp['bn'], stp['conv'].out_channels
        if (stp['tail'] is not None or bn is None or not bn.training or bn.momentum is None or not 0.0 <= stp['slope'] <= 1.0
                or stp['conv'].in_channels != C):
            return False
        if L.kp_conv_pair_supported(2 * N, H, W, C, K, 1 if idx == 0 else 0, code) != 1:
            return False
        if idx > 0 and L.kp_conv_bnin_supported(N, H, W, C, K, code) != 1:
            return False
        rows = L.kp_conv_fwd_stats_rows(N, H + 2, W + 2, C, K, 3, code)
        if L.kp_bn_finalize_pair_supported(rows, K) != 1:   # (the library's own rule: nothing about the finalize is restated here)
            return False
        C = K
    Kout = plan[-1]['conv'].out_channels
    if L.kp_head_pair_supported(2 * N, H, W, C, Kout, int(bool(out_t)), code) != 1:
        return False
    return (L.kp_head_t_supported if out_t else L.kp_head_bnin_supported)(N, H, W, C, Kout, code) == 1


def _materialise(fin, N, H, W, C, pad, tdt, code, st):
    """The activation a fused consumer would have formed on its load path, written out: kp_bn_act_fwd of the saved conv output
    (`fin`: see UnitFunction.forward) with a replicate border of `pad`."""
    a = torch.empty(N, H + 2 * pad, W + 2 * pad, C, device=fin['y'].device, dtype=tdt)
    call('kp_bn_act_fwd', ptr(fin['y']), *fin['bn'], fin['slope'], TAILS[None], ptr(a), pad, None, N, H, W, C, code, st)
    return a


def _wgrad_layer(sv, dy, gw, parts, N, H, W, C, K, dp, tdt, code, dev):
    """Weight gradient of one 3x3 layer on torch's current stream: kp_conv_wgrad (or its fused-BatchNorm-input form) into `parts`
    partial buffers, then kp_wgrad_unpack adds them into the arena view `gw` in a fixed order.  (A module-level function so that
    tools/probes can time a step without it: keypoints_amd._probe.ablate_wgrad.)"""
    wst = stream_ptr()
    if sv['fin'] is None:
        # matrix-core channels on a map that is not a multiple of the tap-reuse kernel's pixel block (no BASELINE shape): zero-padded
        # copies of both operands take that kernel instead of the per-tap fallback -- exact (dy is zero there), ~15x faster
        import ctypes
        hp, wp = ctypes.c_int(0), ctypes.c_int(0)
        if _lib.lib().kp_conv_wgrad_pad_hw(N, H, W, C, K, dp, code, ctypes.byref(hp), ctypes.byref(wp)) == 1:
            Hp, Wp = hp.value, wp.value
            xp = torch.empty(N, Hp + 2, Wp + 2, C, device=dev, dtype=tdt)
            dyp = torch.empty(N, Hp + 2 * dp, Wp + 2 * dp, K, device=dev, dtype=tdt)
            call('kp_pad_hw', ptr(sv['a']), ptr(xp), N, H + 2, W + 2, Hp + 2, Wp + 2, C, code, wst)
            call('kp_pad_hw', ptr(dy), ptr(dyp), N, H + 2 * dp, W + 2 * dp, Hp + 2 * dp, Wp + 2 * dp, K, code, wst)
            parts = _lib.lib().kp_conv_wgrad_partials(N, Hp, Wp, C, K, 3, dp, code)
            scratch = (torch.zeros if parts == 1 else torch.empty)(parts, K, 3, 3, C, device=dev, dtype=torch.float32)
            call('kp_conv_wgrad', ptr(xp), ptr(dyp), ptr(scratch), N, Hp, Wp, C, K, 3, dp, code, wst)
            call('kp_wgrad_unpack', ptr(scratch), parts, ptr(gw), K, sv['Cl'], C, 3, wst)
            return
    scratch = (torch.zeros if parts == 1 else torch.empty)(parts, K, 3, 3, C, device=dev, dtype=torch.float32)
    fused = False
    if sv['fin'] is not None:   # the input side is read from the previous layer's conv output (see forward)
        f = sv['fin']
        fused = try_call('kp_conv_wgrad_bnin', ptr(f['y']), *f['bn'], f['slope'], ptr(dy), ptr(scratch), N, H, W, C, K, dp, code, wst)
        if not fused:
            # the library declined the fused form NOW (a library knob changed since the forward): materialise the
            # activation the forward skipped and take the plain kernel -- same values bit for bit
            sv['a'] = _materialise(f, N, H, W, C, 1, tdt, code, wst)
            parts2 = _lib.lib().kp_conv_wgrad_partials(N, H, W, C, K, 3, dp, code)
            if parts2 != parts:
                parts = parts2
                scratch = (torch.zeros if parts == 1 else torch.empty)(parts, K, 3, 3, C, device=dev, dtype=torch.float32)
    if not fused:
        call('kp_conv_wgrad', ptr(sv['a']), ptr(dy), ptr(scratch), N, H, W, C, K, 3, dp, code, wst)
    call('kp_wgrad_unpack', ptr(scratch), parts, ptr(gw), K, sv['Cl'], C, 3, wst)


class UnitFunction(torch.autograd.Function):
    """forward(x NCHW fp32) -> out NCHW fp32; parameter gradients are accumulated straight into the
    Unit's GradArena (returned as None to autograd), the input gradient is returned normally."""

    @staticmethod
    def forward(ctx, x, unit, record, fmt, xs, x2, *params):
        """x2 (None, or a second fp32 NCHW input of the same batch and map size): the Unit runs on torch.cat((x, x2), dim=1) --
        KeyNet's decoder input, keynet.py:29 -- without that tensor being formed (kp_nchw2_to_nhwc; the backward splits the input
        gradient, kp_nhwc_grad_to_nchw2).
        xs (None, or the SOURCE frame of a Transporter pair -- same shape as x, no gradient): both frames go through the Unit as
        ONE batch with per-half BatchNorm statistics (`_forward_pair`, the kp_*_pair entry points) and the call returns
        (out_source, out); the caller (Unit.forward_pair) has asked `pair_ok` first.
        fmt = (in_t, out_t): the "T" interchange formats of include/kp_hip.h (16-bit NHWC tensors between Units instead of the
        reference's fp32 NCHW).  in_t: x is the first 3x3 layer's input as that layer reads it, T [N][H+2][W+2][Cin] with its
        replicate border (written by the T form of the keypoint bottleneck), and the returned input gradient is the data
        gradient over that padded domain (the producer folds the border).  out_t: the head writes T [N][H][W][K] and the
        backward takes the output gradient in that form.  The callers (TransporterNet) ask *_t_supported first."""
        require_device(x, x2, *params)
        in_t, out_t = fmt
        if xs is not None:
            if x2 is not None:
                raise RuntimeError('keypoints_amd Unit: a pair call takes one input tensor per frame')
            return UnitFunction._forward_pair(ctx, xs, x, unit, record, out_t)
        plan = unit.plan()
        tdt, code = precision_dtype(unit.precision)
        st = stream_ptr()
        dev = x.device
        first = plan[0]
        pad0 = 1 if first['kind'] == 'conv3' else 0
        C1 = 0
        if in_t:
            if x2 is not None:
                raise RuntimeError('keypoints_amd Unit: a T-format input cannot be concatenated')
            if x.dtype != tdt or pad0 != 1:
                raise RuntimeError('keypoints_amd Unit: a T-format input is the padded input of a first 3x3 layer in the Unit\'s storage type')
            x = x.contiguous()
            N, H, W, Cin = x.shape[0], x.shape[1] - 2, x.shape[2] - 2, x.shape[3]
            Cp0 = _first_cp(plan, Cin, unit.precision)
            if Cp0 != Cin:
                raise RuntimeError('keypoints_amd Unit: a T-format input cannot be channel-padded')
            a = x
        else:
            x = f32c(x)
            N, Cin, H, W = x.shape
            if x2 is not None:
                x2 = f32c(x2)
                if x2.dim() != 4 or x2.shape[0] != N or x2.shape[2:] != x.shape[2:]:
                    raise RuntimeError(f'keypoints_amd Unit: cannot concatenate {tuple(x.shape)} and {tuple(x2.shape)} along the channels')
                C1 = x2.shape[1]
                Cin += C1
            Cp0 = _first_cp(plan, Cin, unit.precision)
            a = torch.empty(N, H + 2 * pad0, W + 2 * pad0, Cp0, device=dev, dtype=tdt)
            if x2 is not None:
                call('kp_nchw2_to_nhwc', ptr(x), Cin - C1, ptr(x2), C1, ptr(a), N, Cp0, H, W, pad0, code, st)
            else:
                call('kp_nchw_to_nhwc', ptr(x), ptr(a), N, Cin, Cp0, H, W, pad0, code, st)
        H0, W0 = H, W
        saved = []
        fin = None   # pending BatchNorm-apply of the previous layer, fused into the next convolution (None: `a` is materialised)
        packs = _pack_unit(plan, Cin, Cp0, record, dev, tdt, code, st)
        C = Cin
        for idx, stp in enumerate(plan[:-1]):
            conv, bn = stp['conv'], stp['bn']
            K = conv.out_channels
            if conv.in_channels != C:
                raise RuntimeError(f'channel mismatch: conv expects {conv.in_channels}, got {C}')
            Cl = C                       # logical input channels of this layer (the conv's)
            C = Cp0 if idx == 0 else C   # physical channels of `a`
            wf, wd = packs[idx]
            training = bn is not None and bn.training
            # the image-sized first layer (Cin <= 3 -> 64, model F's encoders) on the recompute kernels: statistics from one pass
            # over the image, then conv + BatchNorm + activation straight into the next layer's padded input; the conv output is
            # never written (the backward recomputes it too: kp_conv_first_bwd_*)
            if (idx == 0 and fin is None and not in_t and C1 == 0 and Cp0 == Cin and training and bn.momentum is not None
                    and stp['tail'] is None and plan[1]['kind'] == 'conv3' and unit.sync_bn is None and 0.0 <= stp['slope'] <= 1.0
                    and runtime.fuse_first_layer() and not (record and x.requires_grad)
                    and _lib.lib().kp_conv_first_fused_supported(N, H, W, Cin, K, code) == 1):
                srows = _lib.lib().kp_conv_first_fused_rows(N, H, W, Cin, K, code)
                stats = torch.empty(srows, 2, K, device=dev, dtype=torch.float32)
                call('kp_conv_first_stats', ptr(a), ptr(wf), ptr(conv.bias), ptr(stats), N, H, W, Cin, K, code, st)
                mean = torch.empty(K, device=dev, dtype=torch.float32)
                invstd = torch.empty(K, device=dev, dtype=torch.float32)
                track = bn.track_running_stats and bn.running_mean is not None
                ws = torch.empty(128 * K, device=dev, dtype=torch.float64)
                call('kp_bn_finalize', ptr(stats), srows, ptr(ws), ptr(mean), ptr(invstd),
                     ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                     ptr(bn.num_batches_tracked) if track else None, float(bn.momentum), float(bn.eps), N * H * W, K, st)
                a_next = torch.empty(N, H + 2, W + 2, K, device=dev, dtype=tdt)
                call('kp_conv_first_fwd_bnact', ptr(a), ptr(wf), ptr(conv.bias), ptr(mean), ptr(invstd), ptr(bn.weight), ptr(bn.bias),
                     stp['slope'], ptr(a_next), N, H, W, Cin, K, code, st)
                if record:
                    saved.append(dict(a=a, fin=None, y=None, mean=mean, invstd=invstd, argmax=None, wd=wd, wf=wf, first_fused=True,
                                      H=H, W=W, C=C, Cl=Cl, K=K))
                a, fin, C = a_next, None, K
                continue
            y = torch.empty(N, H, W, K, device=dev, dtype=tdt)
            stats = torch.empty((N * H * W + 63) // 64, 2, K, device=dev, dtype=torch.float32) if training else None
            mean = invstd = None
            finalized = False
            if training:
                mean = torch.empty(K, device=dev, dtype=torch.float32)
                invstd = torch.empty(K, device=dev, dtype=torch.float32)
                track = bn.track_running_stats and bn.running_mean is not None
                if bn.momentum is None:
                    # torch: cumulative moving average, factor 1 / num_batches_tracked (after this batch's increment).
                    # The counter lives on the device; reading it here is a host sync, taken only in this
                    # non-default mode (the reference never sets momentum=None).
                    mom = 1.0 / (int(bn.num_batches_tracked.item()) + 1) if track else 0.0
                else:
                    mom = float(bn.momentum)
                ws = torch.empty(128 * K, device=dev, dtype=torch.float64)
                fin_args = (ptr(ws), ptr(mean), ptr(invstd), ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                            ptr(bn.num_batches_tracked) if track else None, mom, float(bn.eps))
            if fin is not None:
                # the previous layer's BatchNorm-apply + activation + replicate pad run on this convolution's staging path
                # (kp_conv_fwd_bnin): its padded activation was never written
                call('kp_conv_fwd_bnin', ptr(fin['y']), *fin['bn'], fin['slope'], ptr(wf), ptr(conv.bias), ptr(y), ptr(stats),
                     N, H, W, C, K, code, st)
            elif training and unit.sync_bn is None and runtime.fold_finalize():
                # convolution + statistics finalize as one call: on the persistent matrix-core kernel the finalize runs in the
                # convolution's tail (kp_conv_fwd_fin, include/kp_hip.h) -- no dependent launch between the convolution and its
                # BatchNorm pass; elsewhere the library launches kp_bn_finalize itself
                call('kp_conv_fwd_fin', ptr(a), ptr(wf), ptr(conv.bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, code,
                     *fin_args, N * H * W, ptr(runtime.ticket_ws(dev)), st)
                finalized = True
            else:
                call('kp_conv_fwd', ptr(a), ptr(wf), ptr(conv.bias), ptr(y), ptr(stats), N, H + 2, W + 2, C, K, 3, code, st)
            srows = _lib.lib().kp_conv_fwd_stats_rows(N, H + 2, W + 2, C, K, 3, code) if training else 0
            if bn is not None:
                if training:
                    count = N * H * W
                    sync = unit.sync_bn
                    if sync is not None:   # batch statistics over the ranks' shards together (world x count samples)
                        stats, srows, count = _allreduce_rows(stats[:srows], sync[0]), 2, count * sync[1]
                    if not finalized:
                        call('kp_bn_finalize', ptr(stats), srows, *fin_args, count, K, st)
                else:
                    # eval mode: the running statistics are constants of the layer (nn.BatchNorm2d.eval()); the backward then has
                    # no batch-mean terms (see `eval_bn` there)
                    # (both are SNAPSHOTS when a backward may follow: the backward recomputes xhat from them, and a train-mode forward
                    #  of the same module or a load_state_dict between this forward and its backward must not change one of the two)
                    mean = bn.running_mean.clone() if record else bn.running_mean
                    invstd = torch.rsqrt(bn.running_var + bn.eps)
            tail = stp['tail']
            Ho, Wo = _tail_dims(H, W, tail)
            nxt = plan[idx + 1]
            out_pad = 1 if nxt['kind'] == 'conv3' else 0
            gamma = bn.weight if bn is not None else None
            beta = bn.bias if bn is not None else None
            # 3x3 -> 3x3 with no pool / upsample between them (vgg.py:30-38) and a consumer kernel that applies BatchNorm +
            # activation + replicate pad while it stages its input: this layer's activation pass is skipped altogether
            fuse_next = (tail is None and nxt['kind'] == 'conv3' and runtime.fuse_bn_input() and 0.0 <= stp['slope'] <= 1.0
                         and _lib.lib().kp_conv_bnin_supported(N, H, W, K, nxt['conv'].out_channels, code) == 1)
            # ... and the last 3x3 layer in front of the 1x1 head likewise (kp_head_fwd_bnin / kp_head_bwd_bnin)
            if (not fuse_next and tail is None and nxt is plan[-1] and runtime.fuse_bn_input() and 0.0 <= stp['slope'] <= 1.0
                    and _lib.lib().kp_head_bnin_supported(N, H, W, K, nxt['conv'].out_channels, code) == 1):
                fuse_next = True
            argmax = None
            if fuse_next:
                a_next = None
                fin_next = dict(y=y, bn=(ptr(mean), ptr(invstd), ptr(gamma), ptr(beta)), slope=stp['slope'],
                                keep=(mean, invstd, gamma, beta))
            else:
                fin_next = None
                a_next = torch.empty(N, Ho + 2 * out_pad, Wo + 2 * out_pad, K, device=dev, dtype=tdt)
                argmax = torch.empty(N, Ho, Wo, K, device=dev, dtype=torch.uint8) if (tail == 'M' and record) else None
                call('kp_bn_act_fwd', ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), stp['slope'], TAILS[tail],
                     ptr(a_next), out_pad, ptr(argmax), N, H, W, K, code, st)
            if record:
                saved.append(dict(a=a, fin=fin, y=y, mean=mean, invstd=invstd, argmax=argmax, wd=wd, H=H, W=W, C=C, Cl=Cl, K=K,
                                  eval_bn=bn is not None and not training))
            a, fin, C, H, W = a_next, fin_next, K, Ho, Wo
        head = plan[-1]
        hconv = head['conv']
        Kout = hconv.out_channels
        if out_t:
            out = torch.empty(N, H, W, Kout, device=dev, dtype=tdt)
            if fin is not None:
                call('kp_head_fwd_t', ptr(fin['y']), 1, *fin['bn'], fin['slope'], ptr(hconv.weight), ptr(hconv.bias), head['slope'],
                     ptr(out), N, H, W, C, Kout, code, st)
            else:
                call('kp_head_fwd_t', ptr(a), 0, None, None, None, None, 0.0, ptr(hconv.weight), ptr(hconv.bias), head['slope'],
                     ptr(out), N, H, W, C, Kout, code, st)
        else:
            out = torch.empty(N, Kout, H, W, device=dev, dtype=torch.float32)
            if fin is not None:   # the head reads the last layer's conv output; its BatchNorm-apply + activation run on the load path
                call('kp_head_fwd_bnin', ptr(fin['y']), *fin['bn'], fin['slope'], ptr(hconv.weight), ptr(hconv.bias), head['slope'],
                     ptr(out), N, H, W, C, Kout, code, st)
            else:
                call('kp_head_fwd', ptr(a), ptr(hconv.weight), ptr(hconv.bias), head['slope'], ptr(out), N, H, W, C, Kout, code, st)
        if record:
            ctx.unit, ctx.saved, ctx.head_in = unit, saved, (a if fin is None else fin)
            ctx.save_for_backward(out)   # an OUTPUT kept as a plain ctx attribute would be a reference cycle
            ctx.dims = (N, Cin, H0, W0, H, W, C, Kout, pad0)
            ctx.cp0 = Cp0
            ctx.c1 = C1
            ctx.fmt = (bool(in_t), bool(out_t))
        return out

    @staticmethod
    def _forward_pair(ctx, xs, x, unit, record, out_t):
        """Source and target frame through the Unit as one batch of 2 N images (kp_conv_fwd_pair / kp_conv_fwd_bnin_pair /
        kp_bn_finalize_pair / kp_head_fwd_bnin_pair, include/kp_hip.h): every launch runs each half exactly as the separate call
        would, so the results -- and the running statistics, updated source first -- are bit-identical to
        `with no_grad: unit(xs)` followed by `unit(x)` (transporter.py:36-39), in half as many launches.  Only the target half
        is recorded for the backward (views of the 2 N tensors)."""
        require_device(xs)
        plan = unit.plan()
        tdt, code = precision_dtype(unit.precision)
        st = stream_ptr()
        dev = x.device
        xs, x = f32c(xs), f32c(x)
        N, Cin, H, W = x.shape
        H0, W0 = H, W
        a = torch.empty(2 * N, H + 2, W + 2, Cin, device=dev, dtype=tdt)
        call('kp_nchw_to_nhwc', ptr(xs), ptr(a), N, Cin, Cin, H, W, 1, code, st)
        call('kp_nchw_to_nhwc', ptr(x), ptr(a[N:]), N, Cin, Cin, H, W, 1, code, st)
        packs = _pack_unit(plan, Cin, Cin, record, dev, tdt, code, st)
        saved = []
        fin2 = fin_t = None   # the previous layer's pending BatchNorm-apply: for both halves (y [2N], [2][K] constants) / the target half
        C = Cin
        for idx, stp in enumerate(plan[:-1]):
            conv, bn = stp['conv'], stp['bn']
            K = conv.out_channels
            wf, wd = packs[idx]
            y = torch.empty(2 * N, H, W, K, device=dev, dtype=tdt)
            rows = _lib.lib().kp_conv_fwd_stats_rows(N, H + 2, W + 2, C, K, 3, code)
            stats = torch.empty(2, rows, 2, K, device=dev, dtype=torch.float32)
            if idx == 0:
                call('kp_conv_fwd_pair', ptr(a), ptr(wf), ptr(conv.bias), ptr(y), ptr(stats), 2 * N, H + 2, W + 2, C, K, code, st)
            else:
                call('kp_conv_fwd_bnin_pair', ptr(fin2['y']), *fin2['bn'], fin2['slope'], ptr(wf), ptr(conv.bias), ptr(y), ptr(stats),
                     2 * N, H, W, C, K, code, st)
            if _lib.last_rows() != rows:   # (the statistics buffer above was sized by the plain kernel's row query: the pair form must agree)
                raise RuntimeError(f'keypoints_amd Unit: the pair convolution left {_lib.last_rows()} statistics rows per half, the '
                                   f'row query said {rows}')
            ms = torch.empty(2, 2, K, device=dev, dtype=torch.float32)   # [mean | invstd][half][K]
            track = bn.track_running_stats and bn.running_mean is not None
            call('kp_bn_finalize_pair', ptr(stats), rows, ptr(ms[0]), ptr(ms[1]), ptr(bn.running_mean) if track else None,
                 ptr(bn.running_var) if track else None, ptr(bn.num_batches_tracked) if track else None, float(bn.momentum),
                 float(bn.eps), N * H * W, K, st)
            mean_t, invstd_t = ms[0, 1], ms[1, 1]
            if record:
                saved.append(dict(a=a[N:] if idx == 0 else None, fin=fin_t, y=y[N:], mean=mean_t, invstd=invstd_t, argmax=None, wd=wd,
                                  H=H, W=W, C=C, Cl=C, K=K))
            fin2 = dict(y=y, bn=(ptr(ms[0]), ptr(ms[1]), ptr(bn.weight), ptr(bn.bias)), slope=stp['slope'], keep=(ms, bn.weight, bn.bias))
            fin_t = dict(y=y[N:], bn=(ptr(mean_t), ptr(invstd_t), ptr(bn.weight), ptr(bn.bias)), slope=stp['slope'],
                         keep=(ms, bn.weight, bn.bias, y))
            C = K
        head = plan[-1]
        hconv = head['conv']
        Kout = hconv.out_channels
        shape = (N, H, W, Kout) if out_t else (N, Kout, H, W)
        odt = tdt if out_t else torch.float32
        out_s = torch.empty(shape, device=dev, dtype=odt)
        out = torch.empty(shape, device=dev, dtype=odt)
        call('kp_head_fwd_bnin_pair', ptr(fin2['y']), *fin2['bn'], fin2['slope'], ptr(hconv.weight), ptr(hconv.bias), head['slope'],
             ptr(out_s), ptr(out), int(bool(out_t)), 2 * N, H, W, C, Kout, code, st)
        ctx.mark_non_differentiable(out_s)
        ctx.set_materialize_grads(False)   # (else autograd fills a zero gradient for out_s at every backward: 58 / 29 MB per encoder at 256 x 84 x 84)
        if record:
            ctx.unit, ctx.saved, ctx.head_in = unit, saved, fin_t
            ctx.save_for_backward(out)
            ctx.dims = (N, Cin, H0, W0, H, W, C, Kout, 1)
            ctx.cp0 = Cin
            ctx.c1 = 0
            ctx.fmt = (False, bool(out_t))
        return out_s, out

    @staticmethod
    def backward(ctx, *gouts):
        gout = gouts[-1]   # (a pair call has two outputs; the source half's is non-differentiable, its gradient None)
        if gout is None:
            raise RuntimeError('keypoints_amd Unit: backward without an output gradient')
        if getattr(ctx, 'saved', None) is None:
            raise RuntimeError('keypoints_amd Unit: backward was already run for this forward; the saved activations are '
                               'released after the first backward (for retain_graph=True call '
                               'keypoints_amd.runtime.set_keep_saved_activations(True) before the forward)')
        unit, saved, a_last = ctx.unit, ctx.saved, ctx.head_in
        (out,) = ctx.saved_tensors
        N, Cin, H0, W0, Hh, Wh, Ch, Kout, pad0 = ctx.dims
        plan = unit.plan()
        tdt, code = precision_dtype(unit.precision)
        st = stream_ptr()
        dev = gout.device
        params = list(unit.parameters())
        arena = unit.grad_arena()
        arena.attach(params)
        cur = torch.cuda.current_stream()
        # the weight-gradient side stream is taken when the first layer asks for it: a Unit whose weight gradients all run inline
        # (Pong-class layers, below) then neither touches nor joins it (two events per Unit backward: 0.4 % of a config-2 step)
        side = None
        side_ok = runtime.multi_stream('wgrad') and not runtime.nested_fork_under_capture()   # (a side stream OF a side stream inside a stream capture crashes hipStreamEndCapture, ROCm 7.0)
        keep = []
        gv = arena.views
        c1 = ctx.c1
        need_x2 = c1 > 0 and ctx.needs_input_grad[5]
        need_x = ctx.needs_input_grad[0] or need_x2

        in_t, out_t = ctx.fmt
        head = plan[-1]
        hconv = head['conv']
        has_body = len(plan) > 1
        g = torch.empty(N, Hh, Wh, Ch, device=dev, dtype=tdt) if (has_body or need_x) else None
        # the workspace makes the head's weight gradient a fixed-order two-stage sum (bitwise reproducible step)
        hws = torch.empty(_lib.lib().kp_head_bwd_ws_floats(N, Hh, Wh, Ch, Kout, code), device=dev, dtype=torch.float32)
        hgrads = (ptr(g), ptr(gv[id(hconv.weight)]), ptr(gv[id(hconv.bias)]) if hconv.bias is not None else None, ptr(hws))
        if out_t:
            # output gradient and saved output are T [N][H][W][K] (see forward); should the library decline NOW (unaligned foreign
            # gradient tensor), both are turned into the fp32 planes of the plain entry points and the code below runs as usual
            gout = gout.to(tdt).contiguous()
            if isinstance(a_last, dict):
                done = try_call('kp_head_bwd_t', ptr(gout), ptr(out), ptr(a_last['y']), 1, *a_last['bn'], a_last['slope'], ptr(hconv.weight),
                                head['slope'], *hgrads, N, Hh, Wh, Ch, Kout, code, st)
            else:
                done = try_call('kp_head_bwd_t', ptr(gout), ptr(out), ptr(a_last), 0, None, None, None, None, 0.0, ptr(hconv.weight),
                                head['slope'], *hgrads, N, Hh, Wh, Ch, Kout, code, st)
            if not done:
                g32 = torch.empty(N, Kout, Hh, Wh, device=dev, dtype=torch.float32)
                o32 = torch.empty(N, Kout, Hh, Wh, device=dev, dtype=torch.float32)
                call('kp_nhwc_to_nchw', ptr(gout), ptr(g32), N, Kout, Hh, Wh, 0, code, st)
                call('kp_nhwc_to_nchw', ptr(out), ptr(o32), N, Kout, Hh, Wh, 0, code, st)
                gout, out, out_t = g32, o32, False
        else:
            gout = f32c(gout)
        if out_t:
            pass
        elif isinstance(a_last, dict):   # the forward's fused form: the head's input is recomputed from the conv output
            if not try_call('kp_head_bwd_bnin', ptr(gout), ptr(out), ptr(a_last['y']), *a_last['bn'], a_last['slope'], ptr(hconv.weight),
                            head['slope'], *hgrads, N, Hh, Wh, Ch, Kout, code, st):
                # declined in the backward although the forward's shape query accepted (unaligned gradient tensors, a knob that
                # changed in between): rebuild the activation the forward never wrote and run the plain kernel
                a_last = _materialise(a_last, N, Hh, Wh, Ch, 0, tdt, code, st)
        if not out_t and not isinstance(a_last, dict):
            if isinstance(ctx.head_in, dict):
                hws = torch.empty(_lib.lib().kp_head_bwd_ws_floats(N, Hh, Wh, Ch, Kout, code), device=dev, dtype=torch.float32)
                hgrads = hgrads[:3] + (ptr(hws),)
            call('kp_head_bwd', ptr(gout), ptr(out), ptr(a_last), ptr(hconv.weight), head['slope'], *hgrads,
                 N, Hh, Wh, Ch, Kout, code, st)
        gp = 0   # border width of g (0 = plain, 1 = padded-domain gradient that still needs folding)
        pending_red = None   # (partial rows, count) of the NEXT layer's BatchNorm-backward sums, left by a fused data gradient
        for idx in range(len(saved) - 1, -1, -1):
            stp, sv = plan[idx], saved[idx]
            conv, bn = stp['conv'], stp['bn']
            H, W, C, K = sv['H'], sv['W'], sv['C'], sv['K']
            tail = stp['tail']
            if tail == 'U':
                g2 = torch.empty(N, H, W, K, device=dev, dtype=tdt)
                call('kp_upsample2_bwd', ptr(g), gp, ptr(g2), N, H, W, K, code, st)
                g, gp, tcode = g2, 0, _lib.TAIL_NONE
            else:
                tcode = TAILS[tail]
            gamma = bn.weight if bn is not None else None
            beta = bn.bias if bn is not None else None
            bn_args = (ptr(sv['mean']), ptr(sv['invstd']), ptr(gamma), ptr(beta))
            if sv.get('first_fused'):
                # the recompute form of the first layer's whole backward: two passes over the incoming gradient, the conv output
                # recomputed from the image in both; dy never exists, the image needs no gradient
                if gp != 1 or pending_red is not None:
                    raise RuntimeError('keypoints_amd Unit: the fused first layer expects the padded-domain gradient of the layer above')
                rows_f = _lib.lib().kp_conv_first_fused_rows(N, H, W, C, K, code)
                partials = torch.empty(rows_f, 2, K, device=dev, dtype=torch.float32)
                fargs = (ptr(g), ptr(sv['a']), ptr(sv['wf']), ptr(conv.bias), *bn_args, stp['slope'])
                call('kp_conv_first_bwd_reduce', *fargs, ptr(partials), N, H, W, C, K, code, st)
                sums = torch.empty(2 * K, device=dev, dtype=torch.float32)
                ws = torch.empty(128 * K, device=dev, dtype=torch.float64)
                call('kp_bn_bwd_finalize', ptr(partials), ptr(ws), rows_f, K, ptr(sums), ptr(gv[id(bn.weight)]), ptr(gv[id(bn.bias)]), st)
                scratch = torch.empty(rows_f, K, 3, 3, C, device=dev, dtype=torch.float32)
                call('kp_conv_first_bwd_wgrad', *fargs, ptr(sums), ptr(scratch), N, H, W, C, K, code, st)
                call('kp_wgrad_unpack', ptr(scratch), rows_f, ptr(gv[id(conv.weight)]), K, sv['Cl'], C, 3, st)
                g = None
                unit._layer_done(idx)
                continue
            if pending_red is not None:   # the layer above left this layer's sums in its data gradient's epilogue (below)
                partials, prows = pending_red
                pending_red = None
            else:
                partials = torch.empty(N * H, 2, K, device=dev, dtype=torch.float32)
                call('kp_bn_act_bwd_reduce', ptr(g), gp, tcode, ptr(sv['argmax']), ptr(sv['y']), *bn_args, stp['slope'],
                     ptr(partials), N, H, W, K, code, st)
                prows = _lib.lib().kp_bn_bwd_rows(gp, tcode, N, H, W, K, code)
            if bn is not None:
                dgam, dbet = ptr(gv[id(bn.weight)]), ptr(gv[id(bn.bias)])
            else:   # no BN: sum of dy is the conv-bias gradient
                dgam, dbet = None, (ptr(gv[id(conv.bias)]) if conv.bias is not None else None)
            sums = torch.empty(2 * K, device=dev, dtype=torch.float32)
            ws = torch.empty(128 * K, device=dev, dtype=torch.float64)
            call('kp_bn_bwd_finalize', ptr(partials), ptr(ws), prows, K, ptr(sums), dgam, dbet, st)
            sync = unit.sync_bn if bn is not None else None
            if sync is not None:
                # dgamma / dbeta above are this rank's sums (the gradient all-reduce averages them like every other
                # gradient); the apply pass needs the sums over ALL ranks, and divides by its own N*H*W: hand it the
                # global sums scaled by 1 / world
                g2rows = _allreduce_rows(partials[:prows], sync[0])
                call('kp_bn_bwd_finalize', ptr(g2rows), ptr(ws), 2, K, ptr(sums), None, None, st)
                sums = sums * (1.0 / sync[1])
            if sv.get('eval_bn'):
                # BatchNorm in eval mode is the affine map z = (y - running_mean) * invstd * gamma + beta with CONSTANT statistics:
                # dy = gamma * invstd * gh without the two batch-mean terms (zero sums for the apply pass), dgamma / dbeta are the
                # sums the finalize just wrote, and the conv bias in front of it does get a gradient: sum dy = gamma * invstd * sum gh
                if conv.bias is not None:
                    gv[id(conv.bias)].add_(bn.weight.detach() * sv['invstd'] * sums[:K])
                sums = torch.zeros_like(sums)
            need_dx = idx > 0 or need_x
            dp = 2 if need_dx else 0
            dy = torch.empty(N, H + 2 * dp, W + 2 * dp, K, device=dev, dtype=tdt)
            call('kp_bn_act_bwd_apply', ptr(g), gp, tcode, ptr(sv['argmax']), ptr(sv['y']), *bn_args, stp['slope'],
                 ptr(sums), ptr(dy), dp, N, H, W, K, code, st)
            # (conv bias in front of train-mode BN: analytically zero gradient -> its arena slot stays 0)
            parts = _lib.lib().kp_conv_wgrad_partials(N, H, W, C, K, 3, dp, code)
            if need_dx:
                gx = torch.empty(N, H + 2, W + 2, C, device=dev, dtype=tdt)
                # the layer below (3x3, no pool / upsample behind it: this layer read its padded activation): the first pass of ITS
                # BatchNorm backward -- sums over gx and its conv output -- in this data gradient's epilogue where the kernel exists
                below = plan[idx - 1] if idx > 0 else None
                if (below is not None and below['kind'] == 'conv3' and below['tail'] is None and runtime.fuse_bn_reduce() and 0.0 <= below['slope'] <= 1.0
                        and not saved[idx - 1].get('first_fused')   # (that layer's sums need ITS conv output, which was never stored)
                        and _lib.lib().kp_conv_dgrad_bnred_supported(N, H, W, K, C, code) == 1):
                    svb, bnb = saved[idx - 1], below['bn']
                    rows_b = _lib.lib().kp_conv_dgrad_bnred_rows(N, H, W, K, C, code)
                    part_b = torch.empty(rows_b, 2, C, device=dev, dtype=torch.float32)
                    call('kp_conv_dgrad_bnred', ptr(dy), ptr(sv['wd']), ptr(gx), ptr(svb['y']), ptr(svb['mean']), ptr(svb['invstd']),
                         ptr(bnb.weight if bnb is not None else None), ptr(bnb.bias if bnb is not None else None), below['slope'],
                         ptr(part_b), N, H, W, K, C, code, st)
                    pending_red = (part_b, rows_b)
                else:
                    call('kp_conv_dgrad', ptr(dy), ptr(sv['wd']), ptr(gx), N, H, W, K, C, code, st)
                g, gp = gx, 1
            else:
                g = None
            # The weight gradient depends on nothing downstream of dy and nothing in this backward depends on it: it goes to
            # the side stream, ordered behind the data gradient (two matrix-core kernels side by side gain nothing), so that
            # it runs beside the HBM-bound BatchNorm passes of the next layer: wgrad_l || reduce / finalize / apply_{l-1}.
            # (Measured: side stream released right after apply_l instead: +0.45 ms per step.)
            # (Pong-class layers, 16 / 32 channels: their weight gradient owns a CU's LDS -- two workgroups of 76 KB -- and nothing
            # of the next layer's passes runs beside it; with the reduce pass folded into the data gradient there is also less to
            # run: on the current stream the config-2 step measures 3.19 vs 3.22 ms, tools/probes/bred_roles_ab.sh.)
            lside = None
            if side_ok and not (SMALL_INLINE and C <= 32 and K <= 32):
                if side is None:
                    side = runtime.side_stream('wgrad')
                lside = side
            if lside is not None:
                lside.wait_stream(cur)
            with torch.cuda.stream(lside) if lside is not None else contextlib.nullcontext():
                _wgrad_layer(sv, dy, gv[id(conv.weight)], parts, N, H, W, C, K, dp, tdt, code, dev)
            if lside is not None:
                # dy and the saved input are read on the side stream: they stay referenced until this stream has waited
                # for it (end of the Unit's backward) -- the caching allocator recycles a block in the order of ITS stream
                # (tensor.record_stream would do the same bookkeeping per block; measured: steps of 160-220 ms at random)
                keep.append((dy, sv['a'], sv['fin']))
                if unit.layer_done is not None:   # a gradient reducer reads the arena from this stream
                    cur.wait_stream(side)
            unit._layer_done(idx)   # the arena slots of plan steps >= idx (and of the head) are final from here on
        dx = dx2 = None
        if need_x and in_t:
            dx = g   # the data gradient over the padded input domain [N][H+2][W+2][Cin] (T): the producer folds the border
        elif need_x and c1:   # the two inputs of a concatenated call (forward: x2) get their channel ranges of the folded gradient
            dx = torch.empty(N, Cin - c1, H0, W0, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
            dx2 = torch.empty(N, c1, H0, W0, device=dev, dtype=torch.float32) if need_x2 else None
            call('kp_nhwc_grad_to_nchw2', ptr(g), ptr(dx), Cin - c1, ptr(dx2), c1, N, ctx.cp0, H0, W0, gp, code, st)
        elif need_x:
            dx = torch.empty(N, Cin, H0, W0, device=dev, dtype=torch.float32)
            call('kp_nhwc_grad_to_nchw', ptr(g), ptr(dx), N, Cin, ctx.cp0, H0, W0, gp, code, st)
        # release the saved activations now (not when the graph dies): hundreds of MB per Unit
        if not runtime.keep_saved_activations():
            ctx.saved = ctx.head_in = ctx.unit = None
        if side is not None:
            cur.wait_stream(side)
            runtime._used.discard(side)
            keep.clear()
        runtime.join_after_backward()   # a backward that runs on a side stream itself is joined when the pass ends
        unit._after_backward()
        return (dx, None, None, None, None, dx2) + (None,) * len(params)
