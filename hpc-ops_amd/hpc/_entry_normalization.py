"""torch.ops.hpc.fused_rmsnorm_with_scale — schema + checks + output allocation.

Mirror of reference src/normalization/entry.cc:20-65 (fused_rmsnorm_with_scale_entry and its
TORCH_LIBRARY_FRAGMENT); the compute is hpc_fused_rmsnorm_with_scale_async in libhpc_amd.so.
"""
import torch

from . import _C

_C.torch_lib.define(
    "fused_rmsnorm_with_scale(Tensor input, Tensor weight, Tensor scale, float eps, bool "
    "is_moe) -> (Tensor, Tensor, Tensor)"
)


def _fused_rmsnorm_with_scale_entry(input, weight, scale, eps, is_moe):
    _C.require(
        input.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16,
        "input and weight must be bfloat16.",
    )
    _C.require(input.is_contiguous() and weight.is_contiguous(), "input/weight must be contiguous")
    _C.require(scale.dtype == torch.float32, "scale must be float32")
    _C.require(scale.numel() >= (2 if is_moe else 1), "scale has too few elements")
    output = torch.empty_like(input, dtype=torch.float8_e4m3fn)
    # the reference allocates all three outputs unconditionally (entry.cc:29-31)
    output_fp32 = torch.empty_like(input, dtype=torch.float32)
    output_scale2 = torch.empty_like(input, dtype=torch.float8_e4m3fn)
    hidden = input.size(-1)
    batch = input.numel() // hidden if hidden else 0
    code = _C.lib.hpc_fused_rmsnorm_with_scale_async(
        _C.ptr(input),
        _C.ptr(weight),
        _C.ptr(output),
        _C.ptr(output_fp32) if is_moe else None,
        _C.ptr(output_scale2) if is_moe else None,
        _C.ptr(scale),
        float(eps),
        batch,
        hidden,
        1 if is_moe else 0,
        _C.stream_of(input),
    )
    _C.check(code, "fused_rmsnorm_with_scale_async")
    return output, output_fp32, output_scale2


_C.torch_lib.impl("fused_rmsnorm_with_scale", _fused_rmsnorm_with_scale_entry, "CUDA")
