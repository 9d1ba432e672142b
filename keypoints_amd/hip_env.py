"""The HIP runtime's own settings that matter to a launch-bound train loop -- set on the APPLICATION's behalf, before HIP initialises.

This is the only module of the package that touches the process environment, and it only WRITES (a variable the HIP runtime reads at
its initialisation); the package itself reads no environment variable (tests/test_host_cpu.py).  Nothing in the package calls it:
bench.py and a train script do.
"""
import os
import warnings

import torch

KERNARG_POOL_BYTES = 64 << 20


def configure(kernarg_pool_bytes: int = KERNARG_POOL_BYTES) -> bool:
    """Call BEFORE the first HIP call of the process (before anything touches `torch.cuda`): enlarges the HIP runtime's kernel-argument
    pool (`HSA_KERNARG_POOL_SIZE`, ROCclr; the default is small enough that a launch-bound train loop wraps it every 15-35 steps).
    Every wrap blocks the launching thread for 8-16 ms (33 ms seen once) while the runtime waits for the device; as long as the host is
    that far ahead of the device the queue absorbs it, otherwise the GPU sits idle for the difference -- the 8-23 ms steps inside the
    2.5 ms steps of the Pong-class configuration (tools/cfg2_stall.py, profiles/r06_cfg2_stall.txt: 16 MiB removed every one of them
    in 4 500 steps; 64 MiB makes a wrap a once-per-thousands-of-steps event).  A value the environment already holds is respected.
    Returns True if the variable is in place for the runtime's initialisation, False (with a warning) if HIP was already initialised."""
    if torch.cuda.is_initialized():
        if 'HSA_KERNARG_POOL_SIZE' not in os.environ:
            warnings.warn('keypoints_amd.hip_env.configure: HIP is already initialised; export '
                          f'HSA_KERNARG_POOL_SIZE={int(kernarg_pool_bytes)} before starting the process instead')
            return False
        return True
    os.environ.setdefault('HSA_KERNARG_POOL_SIZE', str(int(kernarg_pool_bytes)))
    return True
