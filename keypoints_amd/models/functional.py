"""Keypoint-bottleneck ops with the reference's names and signatures
(/root/reference/keypoints/models/functional.py), each backed by a HIP kernel through
the C ABI (include/kp_hip.h).  Inputs must live on a HIP device.

    spacial_logsoftmax / spacial_softmax  -> functional.py:27-44
    gaussian_like_function                -> functional.py:56-63
    point_map                             -> functional.py:66-72
    transport_max                         -> transporter.py:57-60 (combine_method='max')
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
        return k, p, q

    @staticmethod
    def backward(ctx, gk, gp, gq):
        k, p, q = ctx.saved_tensors
        B, K, H, W = ctx.shape
        dh = torch.empty(B, K, H, W, device=k.device, dtype=torch.float32)
        call('kp_spatial_softmax_bwd', ptr(f32c(gk)), ptr(k), ptr(p), ptr(q), ptr(dh), B * K, H, W, stream_ptr())
        return dh, None


def spacial_logsoftmax(heatmap, probs=False):
    k, p, q = _SpatialSoftmax.apply(heatmap, True)
    return (k, (p, q)) if probs else k


def spacial_softmax(heatmap, probs=False):
    k, p, q = _SpatialSoftmax.apply(heatmap, False)
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
        return phi, mask_s, mask_t

    @staticmethod
    def backward(ctx, gphi, _gs, _gt):
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
    return _TransportMax.apply(phi_s, phi_t, m_s, m_t)
