"""Keypoint-bottleneck ops with the reference's names and signatures
(/root/reference/keypoints/models/functional.py), each backed by a HIP kernel through
the C ABI (include/kp_hip.h).  Inputs must live on a HIP device.

    spacial_logsoftmax / spacial_softmax  -> functional.py:27-44
    gaussian_like_function                -> functional.py:56-63
    point_map                             -> functional.py:66-72
    transport_max                         -> transporter.py:57-60 (combine_method='max')
    keypoint_bottleneck                   -> all of the above for both branches of TransporterNet.forward
                                             (transporter.py:27-32,57-60) in one launch per direction
"""
from __future__ import annotations

import torch

from .._lib import call, ptr
from ..runtime import f32c, require_device, stream_ptr


class _SpatialSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, heatmap, log_domain):
        require_device(heatmap)
        h = f32c(heatmap)
        B, K, H, W = h.shape
        k = torch.empty(B, K, 2, device=h.device, dtype=torch.float32)
        p = torch.empty(B, K, H, device=h.device, dtype=torch.float32)
        q = torch.empty(B, K, W, device=h.device, dtype=torch.float32)
        call('kp_spatial_softmax_fwd', ptr(h), ptr(k), ptr(p), ptr(q), B * K, H, W, int(log_domain), stream_ptr())
        ctx.save_for_backward(k, p, q)
        ctx.shape = (B, K, H, W)
        ctx.mark_non_differentiable(p, q)   # marginals are display-only in the reference (utils.py:142)
        ctx.set_materialize_grads(False)    # (no zero-filled gradients for them at every backward; k's is the only one there can be)
        return k, p, q

    @staticmethod
    def backward(ctx, gk, gp, gq):
        if gk is None:   # (set_materialize_grads(False): an undefined incoming gradient is zero, and so is this op's)
            return None, None
        k, p, q = ctx.saved_tensors
        B, K, H, W = ctx.shape
        dh = torch.empty(B, K, H, W, device=k.device, dtype=torch.float32)
        call('kp_spatial_softmax_bwd', ptr(f32c(gk)), ptr(k), ptr(p), ptr(q), ptr(dh), B * K, H, W, stream_ptr())
        return dh, None


def spacial_logsoftmax(heatmap, probs=False):
    k, p, q = _SpatialSoftmax.apply(unlazy(heatmap), True)
    return (k, (p, q)) if probs else k


def spacial_softmax(heatmap, probs=False):
    k, p, q = _SpatialSoftmax.apply(unlazy(heatmap), False)
    return (k, (p, q)) if probs else k


class _GaussianLike(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kp, height, width, sigma, eps):
        require_device(kp)
        k = f32c(kp)
        B, K, _ = k.shape
        m = torch.empty(B, K, height, width, device=k.device, dtype=torch.float32)
        call('kp_gaussian_fwd', ptr(k), ptr(m), B * K, height, width, float(sigma), float(eps), stream_ptr())
        ctx.save_for_backward(k)
        ctx.args = (height, width, float(sigma), float(eps))
        return m

    @staticmethod
    def backward(ctx, gm):
        (k,) = ctx.saved_tensors
        height, width, sigma, eps = ctx.args
        dk = torch.empty_like(k)
        call('kp_gaussian_bwd', ptr(f32c(gm)), ptr(k), ptr(dk), k.shape[0] * k.shape[1], height, width, sigma, eps,
             stream_ptr())
        return dk, None, None, None, None


def gaussian_like_function(kp, height, width, sigma=0.1, eps=1e-6):
    return _GaussianLike.apply(kp, int(height), int(width), sigma, eps)


def point_map(kp, h, w):
    """Negative squared distance to each keypoint (used by the 'pretrained_network' mapper; no grad,
    the reference only calls it under no_grad -- transporter.py:90-92)."""
    require_device(kp)
    k = f32c(kp.detach())
    B, K, _ = k.shape
    m = torch.empty(B, K, h, w, device=k.device, dtype=torch.float32)
    call('kp_point_map_fwd', ptr(k), ptr(m), B * K, int(h), int(w), stream_ptr())
    return m


class _TransportMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, phi_s, phi_t, m_s, m_t):
        require_device(phi_s, phi_t, m_s, m_t)
        phi_s, phi_t, m_s, m_t = (f32c(t) for t in (phi_s, phi_t, m_s, m_t))
        B, C, H, W = phi_s.shape
        K = m_s.shape[1]
        dev = phi_s.device
        phi = torch.empty_like(phi_s)
        mask_s = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32)
        mask_t = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32)
        arg_s = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
        arg_t = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
        call('kp_transport_max_fwd', ptr(phi_s), ptr(phi_t), ptr(m_s), ptr(m_t), ptr(phi), ptr(mask_s), ptr(mask_t),
             ptr(arg_s), ptr(arg_t), B, C, K, H, W, stream_ptr())
        ctx.save_for_backward(phi_s, phi_t, mask_s, mask_t, arg_s, arg_t)
        ctx.K = K
        ctx.mark_non_differentiable(mask_s, mask_t)
        ctx.set_materialize_grads(False)
        return phi, mask_s, mask_t

    @staticmethod
    def backward(ctx, gphi, _gs, _gt):
        if gphi is None:   # (undefined incoming gradient = zero)
            return None, None, None, None
        phi_s, phi_t, mask_s, mask_t, arg_s, arg_t = ctx.saved_tensors
        B, C, H, W = phi_s.shape
        K = ctx.K
        need = ctx.needs_input_grad
        d_phi_s = torch.empty_like(phi_s) if need[0] else None
        d_phi_t = torch.empty_like(phi_t) if need[1] else None
        d_m_s = torch.empty(B, K, H, W, device=phi_s.device, dtype=torch.float32) if need[2] else None
        d_m_t = torch.empty(B, K, H, W, device=phi_s.device, dtype=torch.float32) if need[3] else None
        call('kp_transport_max_bwd', ptr(f32c(gphi)), ptr(phi_s), ptr(phi_t), ptr(mask_s), ptr(mask_t), ptr(arg_s),
             ptr(arg_t), ptr(d_phi_s), ptr(d_phi_t), ptr(d_m_s), ptr(d_m_t), B, C, K, H, W, stream_ptr())
        return d_phi_s, d_phi_t, d_m_s, d_m_t


def transport_max(phi_s, phi_t, m_s, m_t):
    """phi_s*(1-max_k m_s)*(1-max_k m_t) + phi_t*max_k m_t -> (phi, mask_s, mask_t)."""
    return _TransportMax.apply(unlazy(phi_s), unlazy(phi_t), unlazy(m_s), unlazy(m_t))


class _TransportMode(torch.autograd.Function):
    """kp_transport_fwd / _bwd: 'sum_and_clamp', 'pretrained_network' and 'loop' (transporter.py:41-55)."""

    @staticmethod
    def forward(ctx, phi_s, phi_t, m_s, m_t, mode):
        from .._lib import COMBINE
        require_device(phi_s, phi_t, m_s, m_t)
        phi_s, phi_t, m_s, m_t = (f32c(t) for t in (phi_s, phi_t, m_s, m_t))
        B, C, H, W = phi_s.shape
        K = m_s.shape[1]
        if m_s.shape != (B, K, H, W) or m_t.shape != m_s.shape or phi_t.shape != phi_s.shape:
            raise RuntimeError('transport: feature maps and keypoint maps must share batch and spatial size')
        phi = torch.empty_like(phi_s)
        mask_s = torch.empty(B, 1, H, W, device=phi_s.device, dtype=torch.float32)
        mask_t = torch.empty(B, 1, H, W, device=phi_s.device, dtype=torch.float32)
        call('kp_transport_fwd', COMBINE[mode], ptr(phi_s), ptr(phi_t), ptr(m_s), ptr(m_t), ptr(phi), ptr(mask_s),
             ptr(mask_t), B, C, K, H, W, stream_ptr())
        ctx.save_for_backward(phi_s, phi_t, m_s, m_t)
        ctx.mode = COMBINE[mode]
        ctx.mark_non_differentiable(mask_s, mask_t)
        ctx.set_materialize_grads(False)
        return phi, mask_s, mask_t

    @staticmethod
    def backward(ctx, gphi, _gs, _gt):
        if gphi is None:   # (undefined incoming gradient = zero)
            return None, None, None, None, None
        phi_s, phi_t, m_s, m_t = ctx.saved_tensors
        B, C, H, W = phi_s.shape
        K = m_s.shape[1]
        need = ctx.needs_input_grad
        d_phi_s = torch.empty_like(phi_s) if need[0] else None
        d_phi_t = torch.empty_like(phi_t) if need[1] else None
        d_m_s = torch.empty_like(m_s) if need[2] else None
        d_m_t = torch.empty_like(m_t) if need[3] else None
        call('kp_transport_bwd', ctx.mode, ptr(f32c(gphi)), ptr(phi_s), ptr(phi_t), ptr(m_s), ptr(m_t), ptr(d_phi_s),
             ptr(d_phi_t), ptr(d_m_s), ptr(d_m_t), B, C, K, H, W, stream_ptr())
        return d_phi_s, d_phi_t, d_m_s, d_m_t, None


def transport(phi_s, phi_t, m_s, m_t, mode):
    """TransporterNet.forward's combine step (transporter.py:41-60) for any `combine_method`
    -> (phi, mask_s, mask_t); gradients flow to all four inputs that require them."""
    phi_s, phi_t, m_s, m_t = unlazy(phi_s), unlazy(phi_t), unlazy(m_s), unlazy(m_t)
    if mode == 'max':
        return _TransportMax.apply(phi_s, phi_t, m_s, m_t)
    if mode not in ('sum_and_clamp', 'pretrained_network', 'loop'):
        raise ValueError(f'unknown combine_method {mode!r}')
    return _TransportMode.apply(phi_s, phi_t, m_s, m_t, mode)


class _KeypointBottleneck(torch.autograd.Function):
    """kp_keypoint_bottleneck_fwd / _bwd: both branches' spacial_logsoftmax + gaussian_like_function + the 'max'
    blend as one kernel; the source branch carries no gradient (transporter.py:36-37)."""

    @staticmethod
    def forward(ctx, heat_s, heat_t, phi_s, phi_t, sigma, eps):
        require_device(heat_s, heat_t, phi_s, phi_t)
        heat_s, heat_t, phi_s, phi_t = (f32c(t) for t in (heat_s, heat_t, phi_s, phi_t))
        B, C, H, W = phi_t.shape
        K = heat_t.shape[1]
        if heat_t.shape != (B, K, H, W) or heat_s.shape != heat_t.shape or phi_s.shape != phi_t.shape:
            raise RuntimeError('keypoint_bottleneck: heat-maps and feature maps must share batch and spatial size')
        dev = phi_t.device
        f32 = dict(device=dev, dtype=torch.float32)
        k = torch.empty(B, K, 2, **f32)
        p = torch.empty(B, K, H, **f32)
        q = torch.empty(B, K, W, **f32)
        m = torch.empty(B, K, H, W, **f32)
        phi = torch.empty_like(phi_t)
        mask_s = torch.empty(B, 1, H, W, **f32)
        mask_t = torch.empty(B, 1, H, W, **f32)
        arg_t = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
        call('kp_keypoint_bottleneck_fwd', ptr(heat_s), ptr(heat_t), ptr(phi_s), ptr(phi_t), ptr(k), ptr(p), ptr(q),
             ptr(m), ptr(phi), ptr(mask_s), ptr(mask_t), ptr(arg_t), B, C, K, H, W, float(sigma), float(eps), stream_ptr())
        ctx.save_for_backward(phi_s, phi_t, mask_s, mask_t, arg_t, k, p, q)
        ctx.args = (float(sigma), float(eps))
        ctx.mark_non_differentiable(p, q, mask_s, mask_t)
        ctx.set_materialize_grads(False)
        return phi, k, p, q, m, mask_s, mask_t

    @staticmethod
    def backward(ctx, gphi, gk, _gp, _gq, gm, _gs, _gt):
        phi_s, phi_t, mask_s, mask_t, arg_t, k, p, q = ctx.saved_tensors
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            raise NotImplementedError('keypoint_bottleneck: the source branch carries no gradient (transporter.py:36-37); '
                                      'compose spacial_logsoftmax / gaussian_like_function / transport_max instead')
        B, C, H, W = phi_t.shape
        K = k.shape[1]
        sigma, eps = ctx.args
        if gphi is None:
            gphi = torch.zeros_like(phi_t)
        d_phi_t = torch.empty_like(phi_t)
        d_heat = torch.empty(B, K, H, W, device=phi_t.device, dtype=torch.float32)
        call('kp_keypoint_bottleneck_bwd', ptr(f32c(gphi)), ptr(f32c(gk)) if gk is not None else None,
             ptr(f32c(gm)) if gm is not None else None, ptr(phi_s), ptr(phi_t), ptr(mask_s), ptr(mask_t), ptr(arg_t),
             ptr(k), ptr(p), ptr(q), ptr(d_phi_t), ptr(d_heat), B, C, K, H, W, sigma, eps, stream_ptr())
        return None, d_heat, None, d_phi_t, None, None


def keypoint_bottleneck(heat_s, heat_t, phi_s, phi_t, sigma=0.1, eps=1e-6):
    """-> (phi, k_t, (p_t, q_t), m_t, mask_s, mask_t): TransporterNet.forward between the encoders and the decoder
    for combine_method='max' (transporter.py:27-32,57-60), one launch forward, one backward."""
    phi, k, p, q, m, mask_s, mask_t = _KeypointBottleneck.apply(heat_s, heat_t, phi_s, phi_t, sigma, eps)
    return phi, k, (p, q), m, mask_s, mask_t


class _KeypointBottleneckT(torch.autograd.Function):
    """kp_keypoint_bottleneck_fwd_t / _bwd_t: _KeypointBottleneck on the 16-bit NHWC ("T") feature maps of its neighbours.
    phi_s / phi_t: the encoder head's output [B][H][W][C] in the Units' storage type (Unit.forward_t(out_t=True)); the blend leaves
    as the decoder's first-layer input [B][H+2][W+2][C] with its replicate border (Unit.forward_t(in_t=True)), whose gradient
    -- the decoder's data gradient over the padded domain -- is folded inside the backward kernel."""

    @staticmethod
    def forward(ctx, heat_s, heat_t, phi_s, phi_t, sigma, eps, code):
        require_device(heat_s, heat_t, phi_s, phi_t)
        heat_s, heat_t = f32c(heat_s), f32c(heat_t)
        phi_s, phi_t = phi_s.contiguous(), phi_t.contiguous()
        B, H, W, C = phi_t.shape
        K = heat_t.shape[1]
        if heat_t.shape != (B, K, H, W) or heat_s.shape != heat_t.shape or phi_s.shape != phi_t.shape or phi_s.dtype != phi_t.dtype:
            raise RuntimeError('keypoint_bottleneck_t: heat-maps and feature maps must share batch and spatial size')
        dev = phi_t.device
        f32 = dict(device=dev, dtype=torch.float32)
        k = torch.empty(B, K, 2, **f32)
        p = torch.empty(B, K, H, **f32)
        q = torch.empty(B, K, W, **f32)
        m = torch.empty(B, K, H, W, **f32)
        a = torch.empty(B, H + 2, W + 2, C, device=dev, dtype=phi_t.dtype)
        mask_s = torch.empty(B, 1, H, W, **f32)
        mask_t = torch.empty(B, 1, H, W, **f32)
        arg_t = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
        call('kp_keypoint_bottleneck_fwd_t', ptr(heat_s), ptr(heat_t), ptr(phi_s), ptr(phi_t), ptr(k), ptr(p), ptr(q), ptr(m),
             ptr(a), ptr(mask_s), ptr(mask_t), ptr(arg_t), B, C, K, H, W, float(sigma), float(eps), code, stream_ptr())
        ctx.save_for_backward(phi_s, phi_t, mask_s, mask_t, arg_t, k, p, q)
        ctx.args = (float(sigma), float(eps), code)
        ctx.mark_non_differentiable(p, q, mask_s, mask_t)
        ctx.set_materialize_grads(False)
        return a, k, p, q, m, mask_s, mask_t

    @staticmethod
    def backward(ctx, ga, gk, _gp, _gq, gm, _gs, _gt):
        phi_s, phi_t, mask_s, mask_t, arg_t, k, p, q = ctx.saved_tensors
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            raise NotImplementedError('keypoint_bottleneck_t: the source branch carries no gradient (transporter.py:36-37)')
        B, H, W, C = phi_t.shape
        K = k.shape[1]
        sigma, eps, code = ctx.args
        if ga is None:
            ga = torch.zeros(B, H + 2, W + 2, C, device=phi_t.device, dtype=phi_t.dtype)
        ga = ga.to(phi_t.dtype).contiguous()
        d_phi_t = torch.empty_like(phi_t)
        d_heat = torch.empty(B, K, H, W, device=phi_t.device, dtype=torch.float32)
        call('kp_keypoint_bottleneck_bwd_t', ptr(ga), ptr(f32c(gk)) if gk is not None else None,
             ptr(f32c(gm)) if gm is not None else None, ptr(phi_s), ptr(phi_t), ptr(mask_s), ptr(mask_t), ptr(arg_t),
             ptr(k), ptr(p), ptr(q), ptr(d_phi_t), ptr(d_heat), B, C, K, H, W, sigma, eps, code, stream_ptr())
        return None, d_heat, None, d_phi_t, None, None, None


def keypoint_bottleneck_t(heat_s, heat_t, phi_s, phi_t, code, sigma=0.1, eps=1e-6):
    """keypoint_bottleneck between T tensors -> (a, k_t, (p_t, q_t), m_t, mask_s, mask_t); `a` = the blend as the decoder's padded
    16-bit NHWC input.  `code`: the storage type's KP_* dtype code."""
    a, k, p, q, m, mask_s, mask_t = _KeypointBottleneckT.apply(heat_s, heat_t, phi_s, phi_t, sigma, eps, code)
    return a, k, (p, q), m, mask_s, mask_t


class _PaddedTToNchw(torch.autograd.Function):
    """kp_nhwc_to_nchw: the interior of a padded T image [B][H+2p][W+2p][C] as fp32 NCHW -- the blended feature map
    TransporterNet.forward RETURNS (transporter.py:64) next to feeding it to the decoder.  Its backward (a loss on the returned
    map: nothing in the reference's train loops) is a zero-bordered layout change in torch."""

    @staticmethod
    def forward(ctx, a, pad, code):
        a = a.contiguous()
        B, Hp, Wp, C = a.shape
        H, W = Hp - 2 * pad, Wp - 2 * pad
        out = torch.empty(B, C, H, W, device=a.device, dtype=torch.float32)
        call('kp_nhwc_to_nchw', ptr(a), ptr(out), B, C, H, W, pad, code, stream_ptr())
        ctx.pad, ctx.dt = pad, a.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        ga = torch.nn.functional.pad(g.permute(0, 2, 3, 1), (0, 0, ctx.pad, ctx.pad, ctx.pad, ctx.pad))
        return ga.to(ctx.dt).contiguous(), None, None


def padded_t_to_nchw(a, pad, code):
    return _PaddedTToNchw.apply(a, int(pad), code)


class LazyNCHW(torch.Tensor):
    """The fp32 NCHW image of a padded 16-bit NHWC tensor, made on FIRST USE.

    `TransporterNet.forward` returns the blended feature map (transporter.py:64) next to feeding it to the decoder; on the T route the
    decoder reads the 16-bit NHWC blend the bottleneck kernel wrote, and the fp32 NCHW copy exists only for the caller.  The
    reference's train loops look at it when they log images (transporter.py:98-108 through utils), never inside a step -- at 256 x 16 x
    84 x 84 the copy is 261 MB of HBM traffic per step for nothing.  This object IS a torch.Tensor to the caller (shape, dtype, device,
    isinstance); the first torch function applied to it (arithmetic, indexing, `.cpu()`, `.data_ptr()`, `backward` ...) runs
    kp_nhwc_to_nchw on the then-current stream and continues on the real tensor.  Autograd: the conversion is recorded when it runs
    (`padded_t_to_nchw` is an autograd Function), so a loss on the returned map reaches the encoders like any other -- as long as it is
    formed before the graph of that forward is freed by a backward, which is when the reference's tensor would have to be used too.
    keypoints_amd's own entry points (`Unit.forward*`, `l2_reconstruction_loss`, the functional ops) unwrap it explicitly.
    Known limit: the wrapper object itself reports `requires_grad == False` (wrapper subclasses cannot carry a grad_fn); every torch
    function applied to it runs on the materialised tensor, which does carry the graph -- code that BRANCHES on `phi.requires_grad`
    before touching the values sees False.  `set_lazy_outputs(False)` returns the plain tensor."""

    @staticmethod
    def __new__(cls, a, pad, code):
        B, Hp, Wp, C = a.shape
        return torch.Tensor._make_wrapper_subclass(cls, (B, C, Hp - 2 * pad, Wp - 2 * pad), dtype=torch.float32, device=a.device,
                                                   requires_grad=False)

    def __init__(self, a, pad, code):
        self._kp_src = (a, int(pad), code)
        self._kp_val = None
        self._kp_detached = False   # _kp_val was made under no_grad although the source carries a graph (see materialize)

    def materialize(self) -> torch.Tensor:
        """The real tensor.  A copy made while autograd was OFF (image logging under torch.no_grad()) of a source that requires grad
        has no graph: it serves the no_grad uses, but the source is kept and the first use with autograd ON makes the copy again, with
        its graph -- a loss on the returned map then reaches the encoders exactly as the reference's `phi_xs` would."""
        if self._kp_val is not None and not (self._kp_detached and torch.is_grad_enabled()):
            return self._kp_val
        a, pad, code = self._kp_src
        val = padded_t_to_nchw(a, pad, code)
        self._kp_val = val
        self._kp_detached = bool(a.requires_grad) and not torch.is_grad_enabled()
        if not self._kp_detached:
            self._kp_src = None
        return val

    # metadata that is known without the values
    _META = {'shape', 'dtype', 'device', 'ndim', 'layout', 'is_cuda', 'is_sparse', 'is_quantized', 'is_meta', 'names'}

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, '__name__', '')
        owner = getattr(func, '__self__', None)
        if name == '__get__' and getattr(owner, '__name__', None) in cls._META:      # t.shape, t.dtype, t.device ...
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        if func in (torch.Tensor.size, torch.Tensor.dim, torch.Tensor.numel, torch.Tensor.is_floating_point, torch.Tensor.get_device,
                    torch.Tensor.element_size, torch.Tensor.is_complex):
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        import torch.utils._pytree as pytree
        unwrap = lambda t: t.materialize() if isinstance(t, LazyNCHW) else t
        with torch._C.DisableTorchFunctionSubclass():
            return func(*pytree.tree_map(unwrap, args), **pytree.tree_map(unwrap, kwargs))


    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # the net under __torch_function__: an ATen operator reached with the wrapper itself (C++ call sites that skip the Python
        # override) runs on the real tensor as well
        import torch.utils._pytree as pytree
        unwrap = lambda t: t.materialize() if isinstance(t, LazyNCHW) else t
        return func(*pytree.tree_map(unwrap, args), **pytree.tree_map(unwrap, kwargs or {}))


def unlazy(t):
    """A LazyNCHW as its real tensor (anything else unchanged): the entry points of this package call it on their tensor arguments."""
    return t.materialize() if isinstance(t, LazyNCHW) else t


def bottleneck_fusable(H, W, C):
    """Shapes the fused kernels take (include/kp_hip.h): everything else composes the piecewise kernels."""
    if H > 256 or W > 256:
        return False
    nq = (H * W) // 4 if (H * W) % 4 == 0 else H * W
    G = min(C, max(1, 1024 // max(1, nq)))
    return G * H * W <= 16384
