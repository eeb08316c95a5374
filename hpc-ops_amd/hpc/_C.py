"""Loader for libhpc_amd.so — the C-ABI boundary (include/hpc_amd.h).

Plays the role of the reference's `hpc/_C.abi3.so` + `torch.ops.load_library`
(reference hpc/__init__.py:43-45): it opens the in-tree shared library with ctypes, declares the
argument types of every `extern "C"` entry point (tests and tools call the C-ABI directly through them), and
loads the C++ host library that registers the ops (reference src/*/entry.cc TORCH_LIBRARY_FRAGMENT blocks).

There is NO fallback: if the library is missing or an entry point is absent, import fails loudly.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p
from pathlib import Path

import torch

# HPC_AMD_DEV=1: the DEVELOPMENT build (libhpc_amd_dev.so, -DHPC_DEV: variant-selection registers and development
# entry points, csrc/hpc_dev.h) instead of the product - for tools/ and the tests marked `dev`.
DEV_BUILD = os.environ.get("HPC_AMD_DEV", "0") == "1"
_LIB_PATH = Path(__file__).resolve().parent / ("libhpc_amd_dev.so" if DEV_BUILD else "libhpc_amd.so")
if not _LIB_PATH.exists():
    raise ImportError(
        f"{_LIB_PATH} not found: build it first with `python hpc-ops_amd/build.py` "
        "(hipcc --offload-arch=gfx950). hpc has no CPU/eager fallback."
    )

lib = ctypes.CDLL(str(_LIB_PATH), mode=ctypes.RTLD_GLOBAL)

P = c_void_p
I = c_int
L = c_int64
F = c_float


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud failure
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


# ---- signatures (keep in the order of include/hpc_amd.h) ------------------------------------
_sig("hpc_version", c_char_p)
_sig("hpc_built_json", c_char_p)
_sig("hpc_get_cu_count", I, I)
if DEV_BUILD:  # the product does not export them
    _sig("hpc_dev_tuning_set", I, I, I)
    _sig("hpc_dev_tuning_get", I, I)
    _sig("hpc_dev_decode_ticket_overruns", I, I)
_sig("hpc_fused_rmsnorm_with_scale_async", I, P, P, P, P, P, P, F, I, I, I, P)
IP = ctypes.POINTER(c_int)
_sig("hpc_attention_decode_num_bins", I, I, I)
_sig("hpc_attention_decode_effective_bins", I, P, I, I, I, I, I)
_sig("hpc_attention_decode_tile_n", I)
_sig("hpc_assign_attention_decode_task_rows", I, IP, I, I, I, I, I, I)
_sig("hpc_assign_attention_decode_task_sync", I, IP, I, I, I, I, I, I, IP, I)
_sig("hpc_assign_attention_decode_task_async", I, IP, IP, I, I, I, I, I, I, P)
_sig("hpc_attention_decode_workspace_bytes", L, I, I, I, I, I)
_sig("hpc_attention_decode_workspace_zero_bytes", L)
_sig("hpc_stream_capture_id", ctypes.c_longlong, P)
_sig("hpc_attention_decode_bf16_async", I, P, P, IP, P, P, P, IP, IP, I, I, I, I, I, I, I, I, I, I, I, I,
     L, L, L, L, L, L, P)
_sig("hpc_attention_decode_fp8_async", I, P, P, IP, P, P, P, IP, IP, P, P, P, I, I, I, I, I, I, I, I, I, I, I,
     I, I, I, L, L, L, L, L, L, L, L, L, P)

_sig("hpc_group_gemm_blockwise_fp8_async", I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, L, L, P, P)
_sig("hpc_moe_count_and_slot_async", I, P, I, I, I, I, I, P, P, P, P, P, P, P)
_sig("hpc_moe_tiles_async", I, P, I, I, P, P, P)
_sig("hpc_moe_gather_blockwise_async", I, P, P, P, P, P, P, I, I, I, I, I, I, I, P, P, P)
_sig("hpc_act_mul_and_blockwise_quant_async", I, P, P, P, P, I, I, L, L, P, P)
_sig("hpc_moe_reduce_async", I, P, P, P, P, P, I, I, I, P)
_sig("hpc_fuse_moe_blockwise_workspace_bytes", L, I, I, I, I, I)
_sig("hpc_fuse_moe_blockwise_async", I, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P)

_sig("hpc_group_gemm_pertensor_fp8_async", I, P, P, P, P, P, P, P, I, I, I, I, I, P, P)
_sig("hpc_act_mul_and_quant_async", I, P, P, P, P, I, I, I, P)
_sig("hpc_scaled_fp8_quant_async", I, P, P, P, L, I, P)
_sig("hpc_moe_gather_rows_async", I, P, P, I, I, I, P, P)
_sig("hpc_fuse_moe_pertensor_async", I, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P)
_sig("hpc_rope_norm_store_kv_async", I, P, P, P, P, P, P, P, P, P, P, P, P, L, L, I, I, I, I, I, I, I, I, I, I, P)
_sig("hpc_rope_norm_store_kv_fp8_async", I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, F, I, L, L,
     I, I, I, I, I, I, I, I, I, I, I, P)
_sig("hpc_gemm_bf16xfp32_splits", I, I, I, I, I)
_sig("hpc_gemm_bf16xfp32_async", I, P, P, P, P, P, P, I, I, I, F, I, I, I, P)
_sig("hpc_topk_router_async", I, IP, P, P, I, I, L, I, I, P)
_sig("hpc_attention_with_kvcache_prefill_fp8_async", I, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I,
     I, I, L, L, L, L, L, L, L, L, L, P)
_sig("hpc_attention_with_kvcache_blocksparse_prefill_fp8_async", I, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I,
     I, I, I, I, I, I, I, I, L, L, L, L, L, L, L, L, L, P)
_sig("hpc_attention_with_kvcache_prefill_bf16_async", I, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I,
     L, L, L, L, L, L, P)
_sig("hpc_attention_prefill_bf16_async", I, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P)
_sig("hpc_masked_act_mul_and_quant_async", I, P, P, P, P, I, I, I, P)
_sig("hpc_masked_act_mul_and_blockwise_quant_async", I, P, P, P, P, I, I, I, P)
_sig("hpc_reformat_x_scale_async", I, P, P, P, P, I, I, I, I, P)
_sig("hpc_sampler_segments", I, I)
_sig("hpc_fused_sampler_workspace_bytes", L, I, I, I)
_sig("hpc_fused_sampler_async", I, P, P, P, I, P, L, P, P, F, P, F, I, P, I, I, P, F, P, I, I, L, I, ctypes.c_uint64, P)
_sig("hpc_fused_sampler_temperature_async", I, P, P, P, I, L, P, F, P, P, I, I, ctypes.c_uint64, P)
PP = ctypes.POINTER(c_void_p)
_sig("hpc_comm_create", I, I, I, I, c_char_p)
_sig("hpc_comm_destroy", I, I)
_sig("hpc_comm_barrier", I, I)
_sig("hpc_comm_allgather", I, I, P, L, P)
_sig("hpc_comm_info", I, I, IP, IP, IP)
_sig("hpc_comm_create_tensor_sync", I, I, L, PP)
_sig("hpc_comm_lookup_peers", I, P, PP, IP)
_sig("hpc_comm_region_bytes_left", L, P)
_sig("hpc_fuse_allreduce_rmsnorm_high_throughput_grid", I, I, I, I)
_sig("hpc_fuse_allreduce_rmsnorm_high_throughput_signal_stride", I, I, I, I)
_sig("hpc_fuse_allreduce_rmsnorm_high_throughput_async", I, PP, PP, PP, P, P, P, F, I, I, I, I, I, I, P)
_sig("hpc_fuse_allreduce_rmsnorm_low_latency_async", I, P, P, P, P, P, P, P, P, F, I, I, I, I, L, P)
_sig("hpc_allreduce_low_latency_async", I, P, P, P, P, P, P, P, P, F, I, I, I, I, L, I, I, P)
_sig("hpc_allreduce_timeouts", I)
_sig("hpc_allreduce_reset_timeouts", I)

# C++ host side (csrc/torch_*.cpp -> hpc/_hpc_torch.so): TORCH_LIBRARY(hpc) with EVERY op of the package registered
# from C++ (+ torch.classes.hpc.MulticastCommunicator), like the reference's src/*/entry.cc -> hpc/_C.abi3.so.  There are
# no Python-side op implementations: the public modules (hpc/*.py) only wrap torch.ops.hpc.* and register fakes.
_SHIM_PATH = Path(__file__).resolve().parent / ("_hpc_torch_dev.so" if DEV_BUILD else "_hpc_torch.so")
if not _SHIM_PATH.exists():
    raise ImportError(f"{_SHIM_PATH} not found: build it first with `python hpc-ops_amd/build.py`")
torch.ops.load_library(str(_SHIM_PATH))

_ERR = {-1: "unsupported configuration", -2: "invalid argument", -3: "HIP launch error",
        -4: "an earlier fused all-reduce timed out waiting for a peer: results since then are undefined, "
            "re-create the communicator handles (hpc_allreduce_reset_timeouts re-arms the entries)"}


def check(code: int, what: str) -> None:
    """Reference convention: TORCH_CHECK(running, "<op> launch failed!") -> RuntimeError."""
    if code != 0:
        raise RuntimeError(f"{what} launch failed! ({_ERR.get(code, code)})")


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream_of(t: torch.Tensor):
    """Current HIP stream of t's device (reference: at::cuda::getCurrentCUDAStream(device)).
    The kernels are launched with the calling thread's CURRENT device, so a tensor that lives on another
    device must not get here silently (its stream belongs to that other device): fail loudly instead - one
    process per GPU is the deployment model, wrap the call in `with torch.cuda.device(t.device)` otherwise."""
    idx = t.device.index
    if idx is not None and idx != torch.cuda.current_device():
        raise RuntimeError(f"hpc: tensor on cuda:{idx} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call under torch.cuda.device(tensor.device)")
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def cu_count(device=None) -> int:
    idx = -1 if device is None else (device.index if device.index is not None else -1)
    n = lib.hpc_get_cu_count(idx)
    if n <= 0:
        raise RuntimeError("hpc_get_cu_count failed (no HIP device?)")
    return n


def require(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(msg)
