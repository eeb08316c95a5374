"""hpc.act — activation + quant surface used by the fused MoE (reference hpc/act.py:7-34, :108-114).
The masked (DeepEP-layout) variants of the reference are outside this hot path."""
import torch
from torch import Tensor

from . import _entry_fuse_moe  # noqa: F401


def act_mul_and_quant(gate_up: Tensor, scale: Tensor, use_bf16_mul: bool = True,
                      output: Tensor = None) -> Tensor:
    """e4m3( silu(gate_up[:, :C]) * gate_up[:, C:] * scale[0] ) for bf16 gate_up [N, 2C]; with
    use_bf16_mul the product is rounded through bf16 like the reference kernel."""
    return torch.ops.hpc.act_mul_and_quant(gate_up, scale, use_bf16_mul, output)


def scaled_fp8_quant(input: Tensor, scale: Tensor = None, output: Tensor = None) -> Tensor:
    """e4m3(input * scale[0]) for a bf16 tensor (scale defaults to 1)."""
    return torch.ops.hpc.scaled_fp8_quant(input, scale, output)
