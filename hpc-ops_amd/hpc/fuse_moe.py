"""hpc.fuse_moe — fused-MoE surface of the reference (hpc/fuse_moe.py:88-262), blockwise FP8 path."""
import torch
from torch import Tensor

from . import _C  # noqa: F401  (loads the libraries that register torch.ops.hpc.*)


def reduce(x: Tensor, topk_pos: Tensor, topk_scale: Tensor, shared_output: Tensor = None) -> Tensor:
    """y[t] = sum_j topk_scale[t, j] * x[topk_pos[t, j]] (+ shared_output[t]); topk_pos < 0 skipped.

    x bf16 [total_num_seq, hidden]; topk_pos int32 / topk_scale float32 [num_seq, num_topk];
    returns bf16 [num_seq, hidden] (reference hpc/fuse_moe.py:88-130).
    """
    return torch.ops.hpc.reduce(x, topk_pos, topk_scale, shared_output)


def fuse_moe_blockwise_fp8(
    x: Tensor,
    x_scale: Tensor,
    gate_up_weight: Tensor,
    gate_up_weight_scale: Tensor,
    down_weight: Tensor,
    down_weight_scale: Tensor,
    topk_ids: Tensor,
    topk_scale: Tensor,
    rank_ep: int,
    num_expert_total: int,
    shared_output: Tensor = None,
) -> Tensor:
    """Run blockwise FP8 FusedMoE (reference hpc/fuse_moe.py:202-229).

    x e4m3 [T, H] with x_scale f32 [T, H/128]; gate_up_weight e4m3 [E, 2I, H] with scales
    [E, 2I/128, pad4(H/128)]; down_weight e4m3 [E, H, I] with scales [E, H/128, pad4(I/128)];
    topk_ids int32 [T, k] (global expert ids; local experts are [rank_ep*E, (rank_ep+1)*E)),
    topk_scale f32 [T, k]; optional shared_output bf16 [T, H].  Returns bf16 [T, H].
    """
    return torch.ops.hpc.fuse_moe_blockwise_fp8(
        x, x_scale, gate_up_weight, gate_up_weight_scale, down_weight, down_weight_scale, topk_ids,
        topk_scale, shared_output, rank_ep, num_expert_total, None,
    )


def fuse_moe_blockwise(
    x: Tensor,
    x_scale: Tensor,
    gate_up_weight: Tensor,
    gate_up_weight_scale: Tensor,
    down_weight: Tensor,
    down_weight_scale: Tensor,
    topk_ids: Tensor,
    topk_scale: Tensor,
    rank_ep: int,
    num_expert_total: int,
    shared_output: Tensor = None,
    output: Tensor = None,
) -> Tensor:
    """Run blockwise FP8 FusedMoE into an optional preallocated output (reference :232-262)."""
    return torch.ops.hpc.fuse_moe_blockwise(
        x, x_scale, gate_up_weight, gate_up_weight_scale, down_weight, down_weight_scale, topk_ids,
        topk_scale, shared_output, rank_ep, num_expert_total, output,
    )


@torch.library.register_fake("hpc::reduce")
def reduce_fake(x, topk_pos, topk_scale, shared_output):
    return torch.empty((topk_pos.size(0), x.size(1)), dtype=torch.bfloat16, device=x.device)


@torch.library.register_fake("hpc::fuse_moe_blockwise_fp8")
def fuse_moe_blockwise_fp8_fake(x, x_scale, gate_up_weight, gate_up_weight_scale, down_weight,
                                down_weight_scale, topk_ids, topk_scale, shared_output, rank_ep,
                                num_expert_total, output):
    return torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)


@torch.library.register_fake("hpc::fuse_moe_blockwise")
def fuse_moe_blockwise_fake(x, x_scale, gate_up_weight, gate_up_weight_scale, down_weight,
                            down_weight_scale, topk_ids, topk_scale, shared_output, rank_ep,
                            num_expert_total, output):
    return torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)


def count_and_gather(x: Tensor, topk_ids: Tensor, num_expert: int, rank_ep: int, intermediate_size: int,
                     num_seq_per_group_avg: int):
    """Sort tokens by expert (reference hpc/fuse_moe.py:8-85).  Returns (gathered x [T*k, H],
    group-GEMM output buffer [T*k, intermediate_size] bf16, topk_pos [T, k], seqlens [E], cu_seqlens
    [E+1], tiles [E], cu_tiles [E+1], and two unused TMA-descriptor placeholders).  Slotting is the
    deterministic arrival order of the reference tests' oracle."""
    return torch.ops.hpc.count_and_gather(x, topk_ids, num_expert, rank_ep, intermediate_size,
                                          num_seq_per_group_avg)


def fuse_moe(x: Tensor, gate_up_weight: Tensor, down_weight: Tensor, gate_up_scale: Tensor,
             down_scale: Tensor, act_and_mul_scale: Tensor, topk_ids: Tensor, topk_scale: Tensor,
             rank_ep: int, num_expert_total: int, use_bf16_mul: bool = True, shared_output: Tensor = None,
             output: Tensor = None) -> Tensor:
    """Run per-tensor FP8 FusedMoE (reference hpc/fuse_moe.py:136-166): one fp32 scale per local
    expert for each GEMM and one activation scale; hidden % 64 == 0, intermediate % 64 == 0."""
    return torch.ops.hpc.fuse_moe(x, gate_up_weight, down_weight, gate_up_scale, down_scale,
                                  act_and_mul_scale, topk_ids, topk_scale, shared_output, rank_ep,
                                  num_expert_total, use_bf16_mul, output)


def fuse_moe_pertensor_fp8(x: Tensor, gate_up_weight: Tensor, down_weight: Tensor, gate_up_scale: Tensor,
                           down_scale: Tensor, act_and_mul_scale: Tensor, topk_ids: Tensor,
                           topk_scale: Tensor, rank_ep: int, num_expert_total: int,
                           use_bf16_mul: bool = True, shared_output: Tensor = None) -> Tensor:
    """Run per-tensor FP8 FusedMoE (reference hpc/fuse_moe.py:169-199)."""
    return torch.ops.hpc.fuse_moe_pertensor_fp8(x, gate_up_weight, down_weight, gate_up_scale, down_scale,
                                                act_and_mul_scale, topk_ids, topk_scale, shared_output,
                                                rank_ep, num_expert_total, use_bf16_mul, None)


@torch.library.register_fake("hpc::fuse_moe")
def fuse_moe_fake(x, gate_up_weight, down_weight, gate_up_scale, down_scale, act_and_mul_scale, topk_ids,
                  topk_scale, shared_output, rank_ep, num_expert_total, use_bf16_mul, output):
    return torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)


@torch.library.register_fake("hpc::fuse_moe_pertensor_fp8")
def fuse_moe_pertensor_fp8_fake(x, gate_up_weight, down_weight, gate_up_scale, down_scale,
                                act_and_mul_scale, topk_ids, topk_scale, shared_output, rank_ep,
                                num_expert_total, use_bf16_mul, output):
    return torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)


@torch.library.register_fake("hpc::count_and_gather")
def count_and_gather_fake(x, topk_ids, num_expert, rank_ep, intermediate_size, num_seq_per_group_avg):
    """nine outputs, as the op's schema and entry say (csrc/torch_moe.cpp::count_and_gather): gathered rows, the gate-up
    output buffer, topk_pos, seqlens, cu_seqlens, tiles, cu_tiles and the two (unused on gfx950) descriptor buffers"""
    rows, dev = topk_ids.size(0) * topk_ids.size(1), x.device
    i32 = dict(dtype=torch.int32, device=dev)
    return (torch.empty((rows, x.size(1)), dtype=x.dtype, device=dev),
            torch.empty((rows, intermediate_size), dtype=torch.bfloat16, device=dev),
            torch.empty(tuple(topk_ids.shape), **i32), torch.empty((num_expert,), **i32), torch.empty((num_expert + 1,), **i32),
            torch.empty((num_expert,), **i32), torch.empty((num_expert + 1,), **i32),
            torch.empty((num_expert * 2, 128), dtype=torch.int8, device=dev),
            torch.empty((num_expert * 2, 128), dtype=torch.int8, device=dev))
