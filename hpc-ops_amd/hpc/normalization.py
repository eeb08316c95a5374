"""hpc.normalization — fused RMSNorm + FP8 quantisation (public surface of reference hpc/normalization.py:6-53)."""
from typing import Tuple, Union

import torch
from torch import Tensor

from . import _C  # noqa: F401  (loads the libraries that register torch.ops.hpc.*)

_F8 = torch.float8_e4m3fn


def fused_rmsnorm_with_scale(a: Tensor, weight: Tensor, eps: float = torch.finfo(torch.float32).eps,
                             scale: Tensor = torch.tensor([1], dtype=torch.float32),
                             is_moe: bool = False) -> Union[Tensor, Tuple[Tensor]]:
    """y = a * rsqrt(mean(a^2) + eps) * weight, quantised to float8_e4m3fn as y / scale[0].

    a: bfloat16 [rows, hidden] (hidden % 8 == 0, <= 16384; the reference instantiates 320 / 4096 / 5120 only);
    weight: bfloat16 [hidden] or [1, hidden]; scale: float32 [1] ([2] with is_moe; moved to a's device).
    Returns the fp8 tensor, or with is_moe the triple (y in fp32, y / scale[0] in fp8, y / scale[1] in fp8).
    """
    dev_scale = scale if scale.device == a.device else scale.to(a.device)
    q0, y32, q1 = torch.ops.hpc.fused_rmsnorm_with_scale(a, weight, dev_scale, eps, is_moe)
    if is_moe:
        return y32, q0, q1
    return q0


@torch.library.register_fake("hpc::fused_rmsnorm_with_scale")
def fused_rmsnorm_with_scale_fake(a, weight, scale, eps, is_moe):
    # schema order (input, weight, scale, eps, is_moe) - the reference's fake swaps eps / scale
    # (hpc/normalization.py:44-53); the op always has three outputs
    return (torch.empty_like(a, dtype=_F8), torch.empty_like(a, dtype=torch.float32), torch.empty_like(a, dtype=_F8))
