"""torch.ops.hpc.fuse_allreduce_rmsnorm_{high_throughput,low_latency} (reference
src/allreduce/entry.cc:14-215: same schemas / checks); compute in csrc/allreduce.hip."""
import ctypes

import torch

from . import _C
from .communicator import lookup_peers

_T = _C.torch_lib
_T.define(  # verbatim: reference src/allreduce/entry.cc:197
    "fuse_allreduce_rmsnorm_high_throughput(Tensor input, Tensor mc_input, Tensor in_residual, Tensor "
    "weight, Tensor signal, int rank, int world_size, int num_max_blocks, float rms_norm_eps, Tensor "
    "output, Tensor mc_output, Tensor out_residual) -> ()"
)
_T.define(  # verbatim: reference src/allreduce/entry.cc:205
    "fuse_allreduce_rmsnorm_low_latency(Tensor input_x, Tensor multicast_x, Tensor data_buffer_ptrs, "
    "Tensor! multinode_x, Tensor buffer_flags, int world_size, int rank, bool rmsnorm_fusion, bool "
    "launch_with_pdl, bool use_two_shot, Tensor! output_x, Tensor! residual_out, Tensor residual_in, "
    "Tensor weight_gamma, float rms_norm_eps) -> ()"
)


def _bf16_contig(t, name):
    _C.require(t.is_cuda and t.is_contiguous(), f"{name} tensor must be a contiguous cuda tensor")
    _C.require(t.dtype == torch.bfloat16, f"{name} must be bfloat16")
    _C.require(t.data_ptr() % 16 == 0, f"{name} must be 16-byte aligned")


_SIGNAL_CACHE = {}


def _signal_ptrs(signal, world_size, rank):
    """Host copy of the device pointer table + the capacity (uint32 words) of this rank's signal pad,
    read once per table (a per-call .cpu() would synchronise the stream and break graph capture).
    The capacity comes from the communicator's registry: the pad runs to the end of the symmetric
    buffer it was carved from (MulticastHandle: 72 * CU count words)."""
    key = (signal.data_ptr(), world_size, rank)
    if key not in _SIGNAL_CACHE:
        ptrs = [int(v) for v in signal.cpu().tolist()[:world_size]]
        left = _C.lib.hpc_comm_region_bytes_left(ctypes.c_void_p(ptrs[rank]))
        _C.require(left >= 4 * world_size, "signal pad is not inside a symmetric buffer of this process (or too small)")
        _SIGNAL_CACHE[key] = (ptrs, int(min(left // 4, 2 ** 31 - 1)))
    return _SIGNAL_CACHE[key]


def _ptr_array(vals):
    arr = (ctypes.c_void_p * 8)()
    for i, v in enumerate(vals):
        arr[i] = v
    return arr


def _ht_entry(x, multicast_x, residual, weight, signal, rank, world_size, num_max_blocks, rms_norm_eps,
              output_x, output_multicast_x, output_residual):
    for t, n in ((x, "x"), (multicast_x, "multicast_x"), (residual, "residual"), (weight, "weight"),
                 (output_x, "output_x"), (output_multicast_x, "output_multicast_x"),
                 (output_residual, "output_residual")):
        _bf16_contig(t, n)
    _C.require(signal.dtype == torch.int64 and signal.numel() >= world_size, "signal must be int64 [world_size]")
    _C.require(1 <= world_size <= 8, "world_size must be in 1..8")
    rows, hidden = x.shape
    _C.require(hidden % 8 == 0 and hidden <= 16384, "hidden_size must be a multiple of 8 and <= 16384")
    _C.require(tuple(residual.shape) == (rows, hidden) and weight.numel() == hidden, "shape mismatch")
    # peers' addresses of this rank's token slice: from the symmetric-memory registry (the reference
    # gets them for free from the NVLS multicast mapping)
    in_ptrs, r_in = lookup_peers(multicast_x)
    out_ptrs, r_out = lookup_peers(output_multicast_x)
    _C.require(len(in_ptrs) == world_size and len(out_ptrs) == world_size and r_in == rank == r_out,
               "multicast views do not belong to a communicator of this world_size / rank")
    sig_ptrs, pad_words = _signal_ptrs(signal, world_size, int(rank))
    rc = _C.lib.hpc_fuse_allreduce_rmsnorm_high_throughput_async(
        _ptr_array(in_ptrs), _ptr_array(out_ptrs), _ptr_array(sig_ptrs), _C.ptr(residual),
        _C.ptr(output_residual), _C.ptr(weight), float(rms_norm_eps), rows, hidden, int(rank),
        int(world_size), int(num_max_blocks), pad_words, _C.stream_of(x))
    _C.check(rc, "fuse_allreduce_rmsnorm_high_throughput_async")


_T.impl("fuse_allreduce_rmsnorm_high_throughput", _ht_entry, "CUDA")


def _ll_entry(input_x, multicast_x, data_buffer_ptrs, multinode_x, buffer_flags, world_size, rank,
              rmsnorm_fusion, launch_with_pdl, use_two_shot, output_x, residual_out, residual_in,
              weight_gamma, rms_norm_eps):
    for t, n in ((input_x, "input_x"), (multinode_x, "multinode_x"), (output_x, "output_x"),
                 (residual_in, "residual_in"), (residual_out, "residual_out"), (weight_gamma, "weight_gamma")):
        _bf16_contig(t, n)
    _C.require(data_buffer_ptrs.dtype == torch.int64 and data_buffer_ptrs.is_cuda
               and data_buffer_ptrs.is_contiguous(), "data_buffer_ptrs must be a contiguous cuda int64 tensor")
    _C.require(buffer_flags.is_cuda and buffer_flags.is_contiguous() and buffer_flags.numel() >= 9
               and buffer_flags.element_size() == 4, "buffer_flags must be 9 x uint32 on the device")
    _C.require(input_x.dim() == 2, "input_x must be 2D [num_tokens, token_dim]")
    _C.require(rmsnorm_fusion and use_two_shot, "only the fused two-shot mode is implemented")
    num_tokens, hidden = input_x.shape
    _C.require(hidden % 8 == 0, "token_dim must be divisible by 8")
    _C.require(tuple(output_x.shape) == (num_tokens, hidden), "output_x shape mismatch")
    _C.require(1 <= world_size <= 64 and 0 <= rank < world_size, "bad world_size / rank")
    _C.require(tuple(residual_in.shape) == (num_tokens, hidden)
               and tuple(residual_out.shape) == (num_tokens, hidden), "residual shape mismatch")
    _C.require(weight_gamma.dim() == 1 and weight_gamma.size(0) == hidden, "weight_gamma shape mismatch")
    rc = _C.lib.hpc_fuse_allreduce_rmsnorm_low_latency_async(
        _C.ptr(output_x), _C.ptr(residual_out), _C.ptr(input_x), _C.ptr(data_buffer_ptrs),
        _C.ptr(multinode_x), _C.ptr(buffer_flags), _C.ptr(residual_in), _C.ptr(weight_gamma),
        float(rms_norm_eps), num_tokens, hidden, int(rank), int(world_size),
        multinode_x.numel() * multinode_x.element_size(), _C.stream_of(input_x))
    _C.check(rc, "fuse_allreduce_rmsnorm_low_latency_async")


_T.impl("fuse_allreduce_rmsnorm_low_latency", _ll_entry, "CUDA")
