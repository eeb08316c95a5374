"""hpc.allreduce — fused AllReduce + residual + RMSNorm surface (reference hpc/allreduce.py:7-200)."""
from typing import Any, Optional, Sequence, Tuple

import torch

from . import _C  # noqa: F401  (loads the libraries that register torch.ops.hpc.*)
from .multicast_handle import MulticastHandle


def fuse_allreduce_rmsnorm_high_throughput(
    x: torch.Tensor,
    multicast_x: torch.Tensor,
    residual: torch.Tensor,
    weight: torch.Tensor,
    rms_norm_eps: float,
    signal: torch.Tensor,
    rank: int,
    world_size: int,
    num_max_blocks: int,
    output_x: Optional[torch.Tensor] = None,
    output_multicast_x: Optional[torch.Tensor] = None,
    output_residual: Optional[torch.Tensor] = None,
) -> None:
    """RMSNorm(AllReduce(x) + residual) * weight, high-throughput two-shot (reference :7-75).

    The caller passes ITS token slice [start:end) of the symmetric input / output buffers (x,
    output_x) together with the matching `get_multimem_buff` views and residual slices; afterwards
    every rank's output buffer holds all rows, output_residual only the local slice.  bf16;
    hidden % 8 == 0, hidden <= 16384; all ranks must pass the same num_max_blocks.
    """
    if output_x is None:
        output_x = x
    if output_multicast_x is None:
        output_multicast_x = multicast_x
    if output_residual is None:
        output_residual = residual
    torch.ops.hpc.fuse_allreduce_rmsnorm_high_throughput(
        x, multicast_x, residual, weight, signal, rank, world_size, num_max_blocks, rms_norm_eps,
        output_x, output_multicast_x, output_residual,
    )


def fuse_allreduce_rmsnorm_low_latency(
    input_x: torch.Tensor,
    multicast_x: torch.Tensor,
    data_buffer_ptrs: torch.Tensor,
    multinode_x: torch.Tensor,
    buffer_flags: torch.Tensor,
    world_size: int,
    rank: int,
    residual_in: torch.Tensor,
    weight_gamma: torch.Tensor,
    rms_norm_eps: float,
    num_max_blocks: int,
    output_x: Optional[torch.Tensor] = None,
    residual_out: Optional[torch.Tensor] = None,
    launch_with_pdl: bool = True,
) -> None:
    """RMSNorm(AllReduce(x) + residual) * weight, low-latency Lamport two-shot (reference :78-123).

    multinode_x: this rank's symmetric workspace [2*ceil(N/ws)*ws*3, H] bf16 pre-filled with
    0x80000000 words; data_buffer_ptrs: device int64 table of every rank's workspace;
    buffer_flags: uint32 [0, 2, bytes_per_slot, 0, 0, 0, 0, 0, 0] advanced on the device.
    num_max_blocks / launch_with_pdl are accepted and ignored (as num_max_blocks is upstream).
    """
    if output_x is None:
        output_x = input_x
    if residual_out is None:
        residual_out = residual_in
    torch.ops.hpc.fuse_allreduce_rmsnorm_low_latency(
        input_x, multicast_x, data_buffer_ptrs, multinode_x, buffer_flags, world_size, rank, True,
        launch_with_pdl, True, output_x, residual_out, residual_in, weight_gamma, rms_norm_eps,
    )


def empty_multimem(
    multicomm,
    *size: Any,
    dtype: Optional[torch.dtype] = None,
    device: Optional[torch.device] = None,
) -> Tuple[torch.Tensor, MulticastHandle]:
    """Allocate a symmetric buffer on every rank (reference hpc/allreduce.py:164-200): returns this
    rank's tensor and the MulticastHandle with the per-rank pointer tables."""
    if len(size) == 1 and isinstance(size[0], Sequence):
        size = tuple(size[0])
    else:
        size = tuple(size)
    if dtype is None:
        dtype = torch.get_default_dtype()
    if device is None:
        device = torch.get_default_device()
    dev_num = (device.index if device.index is not None else 0) if device.type == "cuda" else -1
    assert dev_num == multicomm.GetDeviceId(), (
        f"device(got {dev_num}) of alloc buffer must be same with multicomm(got {multicomm.GetDeviceId()})")
    hdl = MulticastHandle(multicomm, size, dtype)
    return hdl.get_buffer(hdl.rank, size, dtype=dtype), hdl


@torch.library.register_fake("hpc::fuse_allreduce_rmsnorm_high_throughput")
def fuse_allreduce_rmsnorm_high_throughput_fake(x, multicast_x, residual, weight, signal, rank, world_size,
                                                num_max_blocks, rms_norm_eps, output_x, output_multicast_x,
                                                output_residual) -> None:
    return None


@torch.library.register_fake("hpc::fuse_allreduce_rmsnorm_low_latency")
def fuse_allreduce_rmsnorm_low_latency_fake(input_x, multicast_x, data_buffer_ptrs, multinode_x, buffer_flags,
                                            world_size, rank, rmsnorm_fusion, launch_with_pdl, use_two_shot,
                                            output_x, residual_out, residual_in, weight_gamma,
                                            rms_norm_eps) -> None:
    return None
