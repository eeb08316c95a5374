"""RMSNorm(+fp8 quant) oracle — TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates reference tests/test_normalization.py:13-28 (reference_torch_rmsnorm_with_scale,
reference_torch_rmsnorm) for CPU tensors.
"""
import torch


def rmsnorm_fp32(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """reference tests/test_normalization.py:22-28."""
    rms = torch.rsqrt(torch.mean(x.float().pow(2), dim=-1, keepdim=True) + eps)
    y = x * rms
    if weight is not None:
        y = y * weight.float()
    return y


def rmsnorm_with_scale_fp8(x, weight, scale, eps) -> torch.Tensor:
    """reference tests/test_normalization.py:13-19; returns the e4m3 result upcast to bf16."""
    y = rmsnorm_fp32(x, weight, eps)
    inv_scale = 1.0 / scale
    return (y * inv_scale).to(torch.float8_e4m3fn).to(torch.bfloat16)
