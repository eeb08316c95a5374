"""hpc.normalization — same public surface as reference hpc/normalization.py:6-53."""
from typing import Tuple, Union

import torch
from torch import Tensor

from . import _entry_normalization  # noqa: F401  (registers torch.ops.hpc.fused_rmsnorm_with_scale)


def fused_rmsnorm_with_scale(
    a: Tensor,
    weight: Tensor,
    eps: float = torch.finfo(torch.float32).eps,
    scale: Tensor = torch.tensor([1], dtype=torch.float32),
    is_moe: bool = False,
) -> Union[Tensor, Tuple[Tensor]]:
    """RMSNorm(a) * weight, divided by `scale`, emitted as float8_e4m3fn.

    Args:
        a: bfloat16 [batch_size, hidden_states] (hidden % 8 == 0, <= 16384; the reference only
            instantiates 320/4096/5120).
        weight: bfloat16 [hidden_states] (or [1, hidden_states]).
        eps: added to mean(x^2) before rsqrt.
        scale: float32 [1], or [2] when is_moe.
        is_moe: also return the fp32 normalised tensor and a second fp8 tensor (/ scale[1]).
    Returns:
        is_moe: (RMSNorm(a) fp32, RMSNorm(a)/scale[0] fp8, RMSNorm(a)/scale[1] fp8)
        else:   RMSNorm(a)/scale[0] fp8
    """
    if scale.device != a.device:
        scale = scale.to(a.device)
    output_fp8, output_fp32, output_fp8_scale2 = torch.ops.hpc.fused_rmsnorm_with_scale(
        a, weight, scale, eps, is_moe
    )
    return (output_fp32, output_fp8, output_fp8_scale2) if is_moe else output_fp8


@torch.library.register_fake("hpc::fused_rmsnorm_with_scale")
def fused_rmsnorm_with_scale_fake(a, weight, scale, eps, is_moe):
    # argument order follows the op schema (the reference's fake has eps/scale swapped,
    # hpc/normalization.py:44-53); always three outputs, like the real op.
    return (
        torch.empty_like(a, dtype=torch.float8_e4m3fn),
        torch.empty_like(a, dtype=torch.float32),
        torch.empty_like(a, dtype=torch.float8_e4m3fn),
    )
