"""hpc.group_gemm — grouped FP8 GEMM surface (reference hpc/group_gemm.py:134-199, blockwise)."""
import torch
from torch import Tensor

from . import _C  # noqa: F401  (loads the libraries that register torch.ops.hpc.*)


def aligned_size(avg: int) -> int:
    """tileM ladder of the reference (src/fuse_moe/entry.cc:525-543): defines the tile-padded column layout of
    transposed x_scale tensors (group g starts at column cu_tiles[g] * tileM)."""
    for lim, val in ((8, 8), (16, 16), (32, 32), (48, 48), (64, 64), (96, 48), (128, 32), (144, 48)):
        if avg <= lim:
            return val
    return 64


def group_gemm_blockwise_fp8(
    x: Tensor,
    weight: Tensor,
    seqlens: Tensor,
    cu_seqlens: Tensor,
    x_scale: Tensor,
    w_scale: Tensor,
    num_seq_per_group_avg: int = 32,
    output: Tensor = None,
    tma_desc: Tensor = None,
    task_map_workspace: Tensor = None,
) -> Tensor:
    """Grouped GEMM, FP8 operands with 128-block scales (reference hpc/group_gemm.py:134-199).

    x e4m3 [total_seq, K]; weight e4m3 [G, N, K]; seqlens int32 [G]; cu_seqlens int32 [G+1];
    x_scale f32 [K/128, total_seq_pad] in the reference's tile-padded column layout (group g starts
    at column cu_tiles[g]*tileM, tileM from num_seq_per_group_avg); w_scale f32
    [G, N/128, pad4(K/128)].  Returns bf16 [total_seq, N].  tma_desc / task_map_workspace are
    accepted for API compatibility and ignored (no TMA on gfx950).
    """
    return torch.ops.hpc.group_gemm_blockwise_fp8(
        x, weight, seqlens, cu_seqlens, x_scale, w_scale, num_seq_per_group_avg, output, tma_desc,
        task_map_workspace,
    )


@torch.library.register_fake("hpc::group_gemm_blockwise_fp8")
def group_gemm_blockwise_fp8_fake(x, weight, seqlens, cu_seqlens, x_scale, w_scale,
                                  num_seq_per_group_avg, output, tma_desc, task_map_workspace):
    return torch.empty((x.shape[0], weight.shape[1]), dtype=torch.bfloat16, device=x.device)


def group_gemm_pertensor_fp8(x: Tensor, weight: Tensor, seqlens: Tensor, cu_seqlens: Tensor,
                             y_scale: Tensor, num_seq_per_group_avg: int = 32, output: Tensor = None,
                             tma_desc: Tensor = None, task_map_workspace: Tensor = None) -> Tensor:
    """Grouped GEMM, FP8 operands, one fp32 output scale per group (reference hpc/group_gemm.py:51-107):
    y[rows of g] = bf16((x @ weight[g].T) * y_scale[g]).  N % 64 == 0, K % 64 == 0."""
    return torch.ops.hpc.group_gemm_pertensor_fp8(x, weight, seqlens, cu_seqlens, y_scale,
                                                  num_seq_per_group_avg, output, tma_desc, task_map_workspace)


def group_gemm_fp8(x: Tensor, weight: Tensor, seqlens: Tensor, cu_seqlens: Tensor, y_scale: Tensor,
                   num_seq_per_group_avg: int = 32, output: Tensor = None, tma_desc: Tensor = None,
                   task_map_workspace: Tensor = None) -> Tensor:
    """Alias kept by the reference (hpc/group_gemm.py:110-131)."""
    return torch.ops.hpc.group_gemm_fp8(x, weight, seqlens, cu_seqlens, y_scale, num_seq_per_group_avg,
                                        output, tma_desc, task_map_workspace)


def reformat_x_scale(
    x_scale: Tensor,
    seqlens: Tensor,
    cu_seqlens: Tensor,
    num_seq_per_group_avg: int,
    output: Tensor = None,
) -> Tensor:
    """Transpose, tile-pad and compact the activation scales of DeepEP-format inputs into the layout
    group_gemm_blockwise_fp8 reads (reference hpc/group_gemm.py:8-48).

    x_scale f32 [total_seq_pad, K/128] (group g's rows start at cu_seqlens[g], seqlens[g] of them valid);
    returns f32 [K/128, total_seq_pad] where group g starts at column (sum_{j<g} ceil(seqlens[j]/tileM))*tileM,
    tileM = 8/16/32/48/64 from num_seq_per_group_avg.  Padding columns are left as they are."""
    return torch.ops.hpc.reformat_x_scale(x_scale, seqlens, cu_seqlens, output, num_seq_per_group_avg)


@torch.library.register_fake("hpc::reformat_x_scale")
def _reformat_x_scale_fake(x_scale, seqlens, cu_seqlens, out_x_scale, num_seq_per_group_avg):
    if out_x_scale is not None:
        return out_x_scale
    return torch.empty((x_scale.shape[1], x_scale.shape[0]), dtype=x_scale.dtype, device=x_scale.device)


@torch.library.register_fake("hpc::group_gemm_fp8_cp_async")
def _group_gemm_fp8_cp_async_fake(x, weight, y_scale, seqlens, cu_seqlens, tiles, cu_tiles, use_task_map=False):
    return torch.empty((x.shape[0], weight.shape[1]), dtype=torch.bfloat16, device=x.device)


@torch.library.register_fake("hpc::group_gemm_fp8_scatter_cp_async")
def _group_gemm_fp8_scatter_cp_async_fake(x, weight, y_scale, row_indices, seqlens, cu_seqlens, tiles, cu_tiles,
                                          use_task_map=False):
    return torch.empty((row_indices.shape[0], weight.shape[1]), dtype=torch.bfloat16, device=x.device)


@torch.library.register_fake("hpc::group_gemm_fp8")
def _group_gemm_fp8_fake(x, weight, seqlens, cu_seqlens, y_scale, num_seq_per_group_avg, output, tma_desc,
                         task_map_workspace):
    if output is not None:
        return output
    return torch.empty((x.size(0), weight.size(1)), dtype=torch.bfloat16, device=x.device)


@torch.library.register_fake("hpc::group_gemm_pertensor_fp8")
def _group_gemm_pertensor_fp8_fake(x, weight, seqlens, cu_seqlens, y_scale, num_seq_per_group_avg, output, tma_desc,
                                   task_map_workspace):
    if output is not None:
        return output
    return torch.empty((x.size(0), weight.size(1)), dtype=torch.bfloat16, device=x.device)
