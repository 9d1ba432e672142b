"""Measurement switches for tools/probes and bench.py -- NOT part of the product path.

Nothing in keypoints_amd/ reads an environment variable: the engine's A/B choices are plain setters (`keypoints_amd.set_*`) and
module attributes with production defaults.  This module is the one place that turns the `KP_*` variables the probe scripts
export into calls of those setters; `bench.py` and the scripts under `tools/` call `apply_env()` explicitly and print what was set.
One switch here computes WRONG gradients by design (`ablate_wgrad`): it exists to time a step without a kernel family.
"""
from __future__ import annotations

import os
import warnings


def disable_roles(*roles):
    """Side-stream roles ('wgrad', 'chain', 'aux') whose work stays on the parent stream."""
    from . import runtime
    runtime._disabled_roles.update(roles)


def no_prepack():
    """Every Unit packs its weights at its own first call again (instead of one launch per network forward)."""
    from . import unit_engine
    unit_engine.PREPACK = False


def small_wgrad_on_side_stream():
    """The weight gradients of the 16 / 32-channel layers back on the 'wgrad' side stream."""
    from . import unit_engine
    unit_engine.SMALL_INLINE = False


def ablate_wgrad():
    """MEASUREMENT ONLY -- wrong gradients: the 3x3 weight-gradient launches (kernel + unpack) are skipped, their arena slots stay
    zero.  For timing a step without that kernel family (tools/probes/ablate.sh)."""
    from . import unit_engine
    warnings.warn('keypoints_amd._probe.ablate_wgrad(): measurement mode -- the 3x3 weight gradients are NOT computed (their arena '
                  'slots stay zero); do not train with this set', RuntimeWarning)
    unit_engine._wgrad_layer = lambda *a, **k: None


def apply_env(env=None) -> list:
    """Translate the KP_* variables of the probe scripts into the switches above; -> the names that were set (for the record)."""
    from . import runtime
    env = os.environ if env is None else env
    used = []
    roles = [r for r in env.get('KP_NO_ROLES', '').split(',') if r]
    if roles:
        disable_roles(*roles); used.append('KP_NO_ROLES=' + ','.join(roles))
    if env.get('KP_NO_BN_REDUCE_FUSION'):
        runtime.set_fuse_bn_reduce(False); used.append('KP_NO_BN_REDUCE_FUSION')
    if env.get('KP_NO_T_INTERCHANGE'):
        runtime.set_t_interchange(False); used.append('KP_NO_T_INTERCHANGE')
    if env.get('KP_NO_PAIR'):
        runtime.set_pair_calls(False); used.append('KP_NO_PAIR')
    if env.get('KP_NO_PREPACK'):
        no_prepack(); used.append('KP_NO_PREPACK')
    if env.get('KP_SMALL_SIDE'):
        small_wgrad_on_side_stream(); used.append('KP_SMALL_SIDE')
    # library-side knobs (kp_config keys of include/kp_hip.h): the library reads no environment variable either
    for var, key in (('KP_WALK', 5), ('KP_FIN_WIDE', 11), ('KP_DISABLE_MFMA', 12), ('KP_WG16', 13), ('KP_CONV_PATCH', 0)):
        if env.get(var):
            from ._lib import call
            call('kp_config', key, int(env[var], 0)); used.append(f'{var}={env[var]}')
    if env.get('KP_FIRST_ROWS'):   # row kernel of the single-channel first layers (library knob KP_CFG_FIRST_ROWS = 14): 0 off, > 1 workgroups
        from ._lib import call
        call('kp_config', 14, int(env['KP_FIRST_ROWS'])); used.append('KP_FIRST_ROWS=' + env['KP_FIRST_ROWS'])
    if env.get('KP_FIRST_KTN'):   # 32-channel blocks per wavefront in kp_conv_first_fwd_bnact (library knob KP_CFG_FIRST_KTN = 10)
        from ._lib import call
        call('kp_config', 10, int(env['KP_FIRST_KTN'])); used.append('KP_FIRST_KTN=' + env['KP_FIRST_KTN'])
    if env.get('KP_NO_FIN_FOLD'):   # convolution + statistics finalize as two launches again (kp_conv_fwd, kp_bn_finalize)
        runtime.set_fold_finalize(False); used.append('KP_NO_FIN_FOLD')
    if env.get('KP_FIN_NT'):        # threads per workgroup of the one-launch finalize kernels (library knob KP_CFG_FIN_NT = 17): 256 / 1024
        from ._lib import call
        call('kp_config', 17, int(env['KP_FIN_NT'])); used.append('KP_FIN_NT=' + env['KP_FIN_NT'])
    if env.get('KP_STATS_STORE'):   # forward statistics at the store stage (library knob KP_CFG_STATS_STORE = 18; measurement)
        from ._lib import call
        call('kp_config', 18, int(env['KP_STATS_STORE'])); used.append('KP_STATS_STORE=' + env['KP_STATS_STORE'])
    if env.get('KP_BRED_PATCH'):    # BatchNorm-backward sums in the patch data gradient's epilogue (library knob KP_CFG_BRED_PATCH = 9): 0 / 1 / 2
        from ._lib import call
        call('kp_config', 9, int(env['KP_BRED_PATCH'])); used.append('KP_BRED_PATCH=' + env['KP_BRED_PATCH'])
    if env.get('KP_NO_FIRST_FUSE'):
        runtime.set_fuse_first_layer(False); used.append('KP_NO_FIRST_FUSE')
    if env.get('KP_EAGER_BLEND'):
        runtime.set_lazy_outputs(False); used.append('KP_EAGER_BLEND')
    if 'wgrad' in env.get('KP_ABLATE', '').split(','):
        ablate_wgrad(); used.append('KP_ABLATE=wgrad')
    return used
