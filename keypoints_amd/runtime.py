"""Small host-side helpers shared by the op wrappers: device checks, stream handle, precision."""
from __future__ import annotations

import torch

from . import _lib

_DTYPES = {'fp32': (torch.float32, _lib.KP_F32), 'bf16': (torch.bfloat16, _lib.KP_BF16), 'fp16': (torch.float16, _lib.KP_F16)}

# storage type of activations / packed weights inside Units; fp32 = strict-parity mode (apex opt_level O0);
# bf16 / fp16 = mixed precision with fp32 master weights (the O2 role).  fp16 is the reference's own 16-bit type
# (3 more mantissa bits than bf16: 8x tighter parity) and, like there, is trained with loss scaling (keypoints_amd.amp).
_default_precision = 'fp32'


def set_default_precision(name: str) -> None:
    global _default_precision
    if name not in _DTYPES:
        raise ValueError(f'precision must be one of {sorted(_DTYPES)}')
    _default_precision = name


def default_precision() -> str:
    return _default_precision


def precision_dtype(name: str):
    return _DTYPES[name]


def stream_ptr() -> int:
    """hipStream_t of torch's current stream (kernels are enqueued there, never synchronised)."""
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('keypoints_amd ops run on a HIP device only (got a CPU tensor); '
                               'there is no CPU fallback -- move the module and inputs to cuda')


def f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32, contiguous (NCHW) view/copy of a boundary tensor."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
