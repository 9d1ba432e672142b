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


# ---- stream-level concurrency -----------------------------------------------------------------------
# The step alternates matrix-core kernels (power / MFMA bound) with HBM-bound BatchNorm passes; kernels of INDEPENDENT
# chains share the CUs when they are on different HIP streams (whatever fits beside the resident workgroups runs there):
# a Unit's weight gradients go to a side stream of the stream its backward runs on, and the Transporter's keypoint
# encoder runs on a side stream of the feature encoder's.  Results do not depend on it (every kernel is deterministic).
_multi_stream = True
_side_streams = {}     # (device index, parent stream id, role) -> torch.cuda.Stream
_parents = {}          # side stream -> the stream it was forked from
_used = set()          # side streams with work their parent has not waited for yet


def set_multi_stream(on: bool) -> None:
    """Switch the side streams off (every kernel of the step then runs on torch's current stream, in program order)."""
    global _multi_stream
    _multi_stream = bool(on)


# side-stream roles ('wgrad', 'chain', 'aux') that stay on the parent stream (A/B measurements: keypoints_amd._probe.disable_roles)
_disabled_roles = set()


def multi_stream(role: str = None) -> bool:
    # (under stream capture -- the inference extractor's graph, keypoints_amd.graph -- the side streams join the capture
    #  when they wait for the capturing stream and become parallel branches of the graph)
    return _multi_stream and role not in _disabled_roles


def side_stream(role: str) -> torch.cuda.Stream:
    """The side stream `role` of torch's current stream (created on first use)."""
    cur = torch.cuda.current_stream()
    key = (cur.device.index, cur.stream_id, role)
    s = _side_streams.get(key)
    if s is None:
        s = _side_streams[key] = torch.cuda.Stream(device=cur.device)
        _parents[s] = cur
    _used.add(s)
    return s


def nested_fork_under_capture() -> bool:
    """True when torch's current stream is itself a side stream and a stream capture is in progress on it."""
    cur = torch.cuda.current_stream()
    return cur in _parents and torch.cuda.is_current_stream_capturing()


def join_side_streams() -> None:
    """torch's current stream waits for everything enqueued on any side stream so far (optimiser / reducer entry: a no-op
    after a backward, which joins on its own -- see join_after_backward)."""
    if not _used:
        return
    cur = torch.cuda.current_stream()
    for s in list(_used):
        if s.device == cur.device:
            if s != cur:
                cur.wait_stream(s)
            _used.discard(s)


def join_after_backward() -> None:
    """Called at the end of a Unit's backward, on the stream the autograd engine runs it on (= the stream of its forward).
    If that is a side stream, the stream it was forked from waits for it here: by the time `backward()` returns, that
    wait is enqueued, so whatever the caller enqueues next on the forward's stream (`p.grad`, the optimiser, the gradient
    reducer) sees finished arenas.  (An engine callback -- `queue_callback` -- would be the general hook, but it runs on
    the engine's worker thread, whose thread-local current stream is the legacy default stream: waiting THERE on a
    capturing stream aborts the capture of keypoints_amd.graph.)"""
    cur = torch.cuda.current_stream()
    parent = _parents.get(cur)
    if parent is not None:
        parent.wait_stream(cur)
        _used.discard(cur)


_held = {}   # side stream -> tensors its last block handed to the parent stream


class fork:
    """`with fork(role, *inputs) as side:` -- inside, `with side:` runs a block on the side stream `role` of the current
    stream (after everything enqueued so far); leaving the outer block makes the current stream wait for it.
    The caching allocator recycles a block in the order of the stream it was allocated on, so tensors that cross streams
    are kept referenced until the other stream is known to be past them: `inputs` (read by the side block) live in the
    caller's frame until the join; what the side block hands back (`side.outputs(...)`) is held until the NEXT fork of
    this side stream has waited for the parent.  (`tensor.record_stream` is the allocator's own mechanism for this; with
    it the step time jumped to 160-220 ms at random.)
    With the side streams switched off (or under stream capture) `with side:` runs the block in place."""

    def __init__(self, role, *inputs):
        self._role, self._inputs = role, [t for t in inputs if t is not None and t.is_cuda]
        self._stream = None

    def __enter__(self):
        if self._inputs and multi_stream(self._role):
            self._cur = torch.cuda.current_stream()
            self._stream = side_stream(self._role)
            self._stream.wait_stream(self._cur)
            _held.pop(self._stream, None)   # the parent's readers of the previous outputs are behind this wait
        return _SideBlock(self)

    def __exit__(self, *exc):
        if self._stream is not None:
            self._cur.wait_stream(self._stream)
            _used.discard(self._stream)
        self._inputs = None
        return False


class _SideBlock:
    def __init__(self, owner):
        self._o = owner

    def __enter__(self):
        if self._o._stream is not None:
            self._ctx = torch.cuda.stream(self._o._stream)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._o._stream is not None:
            self._ctx.__exit__(*exc)
        return False

    def outputs(self, *tensors):
        if self._o._stream is not None:
            _held[self._o._stream] = tensors


def freeze_gc():
    """Call once after the first train steps of a launch-bound network (Pong-class nets: ~130 launches per 4 ms step).
    The modules, plans, packed weights and optimizer state are long-lived Python objects; every full collection of the
    cyclic garbage collector walks all of them (~40 ms of host time, during which no kernel is launched).  `gc.freeze()`
    moves everything alive now to the permanent generation, so later collections look only at what a step creates."""
    import gc
    gc.collect()
    gc.freeze()


def configure_hip_runtime(**kw) -> bool:
    """`keypoints_amd.hip_env.configure` (the one module of the package that touches the process environment -- it WRITES a variable
    of the HIP runtime on the application's behalf, before HIP initialises; nothing in the package reads one)."""
    from . import hip_env
    return hip_env.configure(**kw)


_FOLD_FINALIZE = False


def fold_finalize() -> bool:
    """OPT-IN (measured slower, docs/experiments/r06.md section 2): a statistics-producing convolution and the BatchNorm finalize behind
    it go to the library as ONE call (kp_conv_fwd_fin): on the persistent matrix-core kernel the finalize then runs in the
    convolution's tail (last workgroup of each output-channel tile) instead of as a dependent launch.  Results equal up to the order
    of the fp64 additions.  Model F step, alternating on one box: 32.71 ms with the separate launch, 32.89 ms folded (the device-scope
    release every workgroup needs before its ticket, and ~3 us of one workgroup summing alone, 56 times per step, cost more than the
    5 us launches they replace); the first build -- a release per THREAD -- cost 1.9 ms."""
    return _FOLD_FINALIZE


def set_fold_finalize(on: bool) -> None:
    global _FOLD_FINALIZE
    _FOLD_FINALIZE = bool(on)


_TICKET_WS = {}


def ticket_ws(device):
    """The KP_FIN_TICKETS zero-initialised counters of kp_conv_fwd_fin for torch's current stream on `device`: every launch leaves them
    zero, so launches on one stream share them; launches on different streams may overlap and get their own.  Created while a HIP
    graph is being captured, the zero-fill is only recorded: such a buffer belongs to that graph and is not cached."""
    key = (device.index, torch.cuda.current_stream(device).stream_id)
    ws = _TICKET_WS.get(key)
    if ws is None:
        ws = torch.zeros(64, device=device, dtype=torch.int32)
        if not torch.cuda.is_current_stream_capturing():
            _TICKET_WS[key] = ws
    return ws


def reserve_memory(factor: float = 1.25, device=None, max_bytes: int = 16 << 30) -> int:
    """Call once after the first one or two train steps of a launch-bound network: hands the caching allocator ONE segment of
    `factor` x the peak it has reserved so far (and drops the unused segments of the warm-up), so that every later request is a
    split of memory the process already owns.  Without it a step whose tensor lifetimes differ slightly from the warm-up's (a
    garbage collection that ran later, a side stream that was a little behind) can need a block no cached segment fits, and
    hipMalloc of a few hundred MB in the middle of a 2.6 ms step costs 10-25 ms (tools/cfg2_stall.py).  Returns the bytes of the
    new segment (0: nothing was needed).  Not needed under a HIP graph (its private pool is fixed at capture).
    For LAUNCH-BOUND workloads only: the peak is the allocator's since the last `torch.cuda.reset_peak_memory_stats()` (call that
    when the workload starts), the segment serves the CURRENT stream's pool (side streams keep their own), and a request above
    `max_bytes` (16 GiB) is declined (returns 0) -- a step that holds tens of GB is not launch-bound, and 1.25 x its peak next to the
    side streams' pools crowds the device (one 3.3 s step at config 5's 136 GB, profiles/r06v_cfg5_bench.json of the first records run)."""
    dev = torch.cuda.current_device() if device is None else device
    torch.cuda.synchronize(dev)
    peak = torch.cuda.max_memory_reserved(dev)
    torch.cuda.empty_cache()
    need = int(factor * peak) - torch.cuda.memory_reserved(dev)
    if need <= 0:
        return 0
    need = (need + (1 << 21) - 1) >> 21 << 21
    if need > max_bytes:
        return 0
    free, _ = torch.cuda.mem_get_info(dev)
    need = min(need, int(0.9 * free))
    if need <= 0:
        return 0
    t = torch.empty(need, dtype=torch.uint8, device=torch.device('cuda', dev) if isinstance(dev, int) else dev)
    del t
    return need


_FUSE_BN_INPUT = True


def fuse_bn_input() -> bool:
    """3x3 -> 3x3 layer pairs run the first layer's BatchNorm-apply + activation inside the second convolution's staging
    (kp_conv_fwd_bnin / kp_conv_wgrad_bnin) where the library has such a kernel for the shape.  Results are bit-identical
    either way; `set_fuse_bn_input(False)` restores the separate pass for A/B measurements and tests."""
    return _FUSE_BN_INPUT


def set_fuse_bn_input(on: bool) -> None:
    global _FUSE_BN_INPUT
    _FUSE_BN_INPUT = bool(on)


_FUSE_BN_REDUCE = True


def fuse_bn_reduce() -> bool:
    """The data gradient of a 3x3 layer leaves the first pass of the BatchNorm backward of the layer below in its epilogue
    (kp_conv_dgrad_bnred) where the library has such a kernel: that pass is not launched.  Same sums up to the order of the fp32
    additions (NOT bit-identical to the separate pass: its own switch, apart from `fuse_bn_input`).
    Measured on config 2 (tools/probes/bred_roles_ab.sh, one box, three rounds): kernel time per step -0.09 ms (five 42 us passes
    gone, the five data gradients 18-30 us longer: they read the conv output the pass read); with the small-channel weight
    gradients on a side stream the step does not move (3.22 -> 3.23 ms: the HBM-bound pass ran beside them, the longer data
    gradient does not), with them on the current stream (unit_engine, the default for these layers) 3.26 -> 3.19 ms.
    `set_fuse_bn_reduce(False)` restores the separate pass."""
    return _FUSE_BN_REDUCE


def set_fuse_bn_reduce(on: bool) -> None:
    global _FUSE_BN_REDUCE
    _FUSE_BN_REDUCE = bool(on)


_T_INTERCHANGE = True


def t_interchange() -> bool:
    """TransporterNet moves the feature maps between the encoder head, the 'max' keypoint bottleneck and the decoder as 16-bit NHWC
    tensors (kp_head_fwd_t / kp_keypoint_bottleneck_fwd_t, include/kp_hip.h) where the library has those kernels (the Pong-class
    nets in bf16 / fp16), instead of the fp32 NCHW tensors of the reference's Unit interface plus two layout kernels.
    `set_t_interchange(False)` restores the fp32 route for A/B measurements and tests."""
    return _T_INTERCHANGE


def set_t_interchange(on: bool) -> None:
    global _T_INTERCHANGE
    _T_INTERCHANGE = bool(on)


_PAIR_CALLS = True


def pair_calls() -> bool:
    """TransporterNet runs the source and the target frame through an encoder as ONE batch with per-half BatchNorm statistics
    (Unit.forward_pair, the kp_*_pair entry points) where the library has those kernels; results are bit-identical to the two
    calls.  `set_pair_calls(False)` restores the two calls for A/B measurements and tests."""
    return _PAIR_CALLS


def set_pair_calls(on: bool) -> None:
    global _PAIR_CALLS
    _PAIR_CALLS = bool(on)


_LAZY_OUTPUTS = True


def lazy_outputs() -> bool:
    """TransporterNet.forward's T route returns the blended feature map (transporter.py:64) as a `functional.LazyNCHW`: a torch.Tensor
    whose fp32 NCHW values are produced by kp_nhwc_to_nchw on first use instead of in every forward (261 MB of traffic per step at
    config 2 for a tensor the train loop only looks at when it logs images).  `set_lazy_outputs(False)` makes the copy eagerly, on a
    side stream beside the decoder, as round 4 did."""
    return _LAZY_OUTPUTS


def set_lazy_outputs(on: bool) -> None:
    global _LAZY_OUTPUTS
    _LAZY_OUTPUTS = bool(on)


_FUSE_FIRST = True


def fuse_first_layer() -> bool:
    """A Unit's image-sized first layer (Cin <= 3 -> 64, 16-bit storage: model F's encoders) runs on the recompute kernels
    (kp_conv_first_stats / _fwd_bnact / _bwd_reduce / _bwd_wgrad, include/kp_hip.h): its conv output and its BatchNorm-backward
    gradient are recomputed from the 48 x smaller image wherever they are needed and never written -- 4 of the 5 passes over the
    layer's [N][H][W][64] tensor per call are gone.  Same values as the plain sequence up to the order of the fp32 additions in the
    statistics / backward sums / weight-gradient partials.  `set_fuse_first_layer(False)` restores the plain sequence (A/B, tests)."""
    return _FUSE_FIRST


def set_fuse_first_layer(on: bool) -> None:
    global _FUSE_FIRST
    _FUSE_FIRST = bool(on)


_KEEP_SAVED = False


def keep_saved_activations() -> bool:
    """`loss.backward(retain_graph=True)`: a Unit's backward releases its saved activations when it returns (hundreds of MB per
    Unit that autograd would otherwise hold until the graph dies) -- a second backward through the same forward then raises.  With
    `set_keep_saved_activations(True)` they stay with the graph and every further backward accumulates into `.grad` as torch's
    does (nothing in a backward writes a saved buffer).  The reference's train loops (/root/reference/transporter.py:85-89,
    keypoints.py:86-90) run one backward per forward: OFF by default."""
    return _KEEP_SAVED


def set_keep_saved_activations(on: bool) -> None:
    global _KEEP_SAVED
    _KEEP_SAVED = bool(on)


