"""hpc.act — activation + quant surface used by the fused MoE (reference hpc/act.py:7-114), including the
masked (DeepEP-layout) variants."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _C  # noqa: F401  (loads the libraries that register torch.ops.hpc.*)


def act_mul_and_quant(gate_up: Tensor, scale: Tensor, use_bf16_mul: bool = True,
                      output: Tensor = None) -> Tensor:
    """e4m3( silu(gate_up[:, :C]) * gate_up[:, C:] * scale[0] ) for bf16 gate_up [N, 2C]; with
    use_bf16_mul the product is rounded through bf16 like the reference kernel."""
    return torch.ops.hpc.act_mul_and_quant(gate_up, scale, use_bf16_mul, output)


def scaled_fp8_quant(input: Tensor, scale: Tensor, output: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Quantise a float32 / float16 / bfloat16 tensor to e4m3 with one per-tensor scale: output = e4m3(input *
    (1 / scale[0])), saturating; returns (output, scale) like the reference (hpc/act.py:108-114,
    src/activation/activation.cu:461-505)."""
    return torch.ops.hpc.scaled_fp8_quant(input, scale, output)


def masked_act_mul_and_quant(gate_up: Tensor, scale: Tensor, num_per_expert: Tensor,
                             output: Optional[Tensor] = None) -> Tensor:
    """DeepEP layout: gate_up bf16 [num_expert * padded_tokens, 2C], only the first num_per_expert[e] rows of
    expert e are valid: e4m3(silu(gate) * up * scale[0]) for those rows, the others are left untouched
    (reference hpc/act.py:37-67)."""
    return torch.ops.hpc.masked_act_mul_and_quant(gate_up, scale, num_per_expert, output)


def masked_act_mul_and_blockwise_quant(gate_up: Tensor, num_per_expert: Tensor, output: Optional[Tensor] = None,
                                       output_scale: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """DeepEP layout, 128-block quantisation: a = silu(gate) * up, scale = amax_128(|a|) / 448 -> float32
    [N, C/128], q = e4m3(a / (scale + 1e-8)) -> [N, C]; valid rows only (reference hpc/act.py:70-105)."""
    return torch.ops.hpc.masked_act_mul_and_blockwise_quant(gate_up, num_per_expert, output, output_scale)


@torch.library.register_fake("hpc::masked_act_mul_and_quant")
def _masked_act_mul_and_quant_fake(input, scale, num_per_expert, output=None):
    if output is not None:
        return output
    return torch.empty(tuple(input.shape[:-1]) + (input.shape[-1] // 2,), dtype=torch.float8_e4m3fn, device=input.device)


@torch.library.register_fake("hpc::masked_act_mul_and_blockwise_quant")
def _masked_act_mul_and_blockwise_quant_fake(input, num_per_expert, output=None, output_scale=None):
    n, c = input.shape[0], input.shape[1] // 2
    out = output if output is not None else torch.empty((n, c), dtype=torch.float8_e4m3fn, device=input.device)
    osc = output_scale if output_scale is not None else torch.empty((n, c // 128), dtype=torch.float32,
                                                                    device=input.device)
    return out, osc


@torch.library.register_fake("hpc::act_mul_and_quant")
def _act_mul_and_quant_fake(input, scale, use_bf16_mul, output):
    if output is not None:
        return output
    return torch.empty(tuple(input.shape[:-1]) + (input.shape[-1] // 2,), dtype=torch.float8_e4m3fn, device=input.device)


@torch.library.register_fake("hpc::scaled_fp8_quant")
def _scaled_fp8_quant_fake(input, scale, output):
    out = output if output is not None else torch.empty_like(input, dtype=torch.float8_e4m3fn)
    sc = scale if scale is not None else torch.empty((1,), dtype=torch.float32, device=input.device)
    return out, sc
