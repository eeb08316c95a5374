"""Planted faults for tests/test_replay_check.py: functions the memory-checked replay harness must reject."""
import ctypes
import os

import torch


def store_past_the_end(x, out):
    """quantises `x` into `out` but tells the kernel that there are 8 more elements than there are: 8 bytes are stored
    past the end of `out` (inside the caching allocator's 512-byte rounding in the calling process - harmless there,
    which is exactly why such a bug survives ordinary tests)"""
    import hpc
    from hpc import _C

    scale = torch.ones(1, dtype=torch.float32, device=x.device)
    rc = _C.lib.hpc_scaled_fp8_quant_async(_C.ptr(out), _C.ptr(x), _C.ptr(scale), x.numel() + 8, 2, _C.stream_of(x))
    assert rc == 0
    return out


def depends_on_the_process(x):
    """a result that differs from run to run (stands for a race / a read of uninitialised scratch)"""
    return x + float(os.getpid() % 97 + 1)


def well_behaved(x, out):
    out.copy_(x * 2)
    return out
