"""torch.ops.hpc.{fuse_moe_blockwise_fp8, fuse_moe_blockwise, reduce, group_gemm_blockwise_fp8}.

Mirror of reference src/fuse_moe/entry.cc:445-684 (fuse_moe_blockwise_entry, reduce_entry) and
src/group_gemm/entry.cc:91-168 (group_gemm_blockwise_fp8_entry): same schemas, checks and output
allocation; compute is in libhpc_amd.so (csrc/fuse_moe.hip, csrc/group_gemm_blockwise.hip).
"""
import torch

from . import _C

_T = _C.torch_lib
_F8 = torch.float8_e4m3fn

_T.define(
    "fuse_moe_blockwise_fp8(Tensor x, Tensor x_scale, Tensor gate_up_weight, Tensor "
    "gate_up_weight_scale, Tensor down_weight, Tensor down_weight_scale, Tensor topk_ids, "
    "Tensor topk_scale, Tensor ? shared_output, int rank_ep, int num_expert_total, Tensor ? "
    "output) -> (Tensor)"
)
_T.define(
    "fuse_moe_blockwise(Tensor x, Tensor x_scale, Tensor gate_up_weight, Tensor "
    "gate_up_weight_scale, Tensor down_weight, Tensor down_weight_scale, Tensor topk_ids, Tensor "
    "topk_scale, Tensor ? shared_output, int rank_ep, int num_expert_total, Tensor ? output) -> (Tensor)"
)
_T.define("reduce(Tensor x, Tensor topk_pos, Tensor topk_scale, Tensor ? shared_output) -> (Tensor)")
_T.define(  # verbatim: reference src/group_gemm/entry.cc:239
    "group_gemm_blockwise_fp8(Tensor x, Tensor weight, Tensor seqlens, Tensor cu_seqlens, Tensor xscale, "
    "Tensor wscale,int num_seq_per_group_avg, Tensor? output, Tensor? tma_desc, Tensor? "
    "task_map_workspace) -> (Tensor)"
)


def aligned_size(avg: int) -> int:
    """tileM ladder of the reference (src/fuse_moe/entry.cc:525-543): defines the tile-padded column
    layout of transposed x_scale tensors."""
    for lim, val in ((8, 8), (16, 16), (32, 32), (48, 48), (64, 64), (96, 48), (128, 32), (144, 48)):
        if avg <= lim:
            return val
    return 64


def _cu_tiles128(seqlens, m, num_group, stream):
    """Scan of ceil(seqlens/128) for the tiled (large-group) GEMM kernel; None keeps the streaming one.
    Returns the TENSOR: the caller keeps it referenced until its GEMM launch has been enqueued (a raw pointer
    into a tensor that died at return would only be safe by grace of the caching allocator's stream ordering)."""
    if m // max(num_group, 1) <= 20:
        return None
    tiles = torch.empty(num_group, dtype=torch.int32, device=seqlens.device)
    cu = torch.empty(num_group + 1, dtype=torch.int32, device=seqlens.device)
    _C.check(_C.lib.hpc_moe_tiles_async(_C.ptr(seqlens), num_group, 128, _C.ptr(tiles), _C.ptr(cu), stream),
             "group_gemm tiles")
    return cu


def _cuda_contig(t, name):
    _C.require(t.is_cuda, f"{name} tensor must be cuda")
    _C.require(t.is_contiguous(), f"{name} tensor must be contiguous")


def _fuse_moe_blockwise_entry(x, x_scale, gate_up_weight, gate_up_weight_scale, down_weight,
                              down_weight_scale, topk_ids, topk_scale, shared_output, rank_ep,
                              num_expert_total, output):
    _C.require(x.dtype == _F8 and gate_up_weight.dtype == _F8 and down_weight.dtype == _F8,
               "x, gate_up_weight and down_weight dtype must be fp8_e4m3")
    _C.require(topk_ids.dtype == torch.int32, "topk_ids dtype must be int32")
    _C.require(gate_up_weight_scale.dtype == torch.float32 and down_weight_scale.dtype == torch.float32
               and topk_scale.dtype == torch.float32 and x_scale.dtype == torch.float32,
               "gate_up_scale, down_scale, x_scale and topk_scale dtype must be float32")
    for t, n in ((x, "x"), (x_scale, "x_scale"), (gate_up_weight, "gate_up_weight"),
                 (gate_up_weight_scale, "gate_up_weight_scale"), (down_weight, "down_weight"),
                 (down_weight_scale, "down_weight_scale"), (topk_ids, "topk_ids"),
                 (topk_scale, "topk_scale")):
        _cuda_contig(t, n)
    _C.require(x.size(0) == topk_ids.size(0), "x and topk_ids must share the same num_tokens")
    _C.require(topk_ids.shape == topk_scale.shape, "topk_ids and topk_scale must share the same shape")
    _C.require(x.size(1) == gate_up_weight.size(2), "x and weight must share the same k")
    _C.require(gate_up_weight.size(0) == down_weight.size(0),
               "gate_up_weight and down_weight must share the same num_expert")
    _C.require(x_scale.size(0) == x.size(0) and x_scale.size(1) == x.size(1) // 128,
               "x_scale must be per 128 blockwise quant")
    _C.require(gate_up_weight_scale.size(1) == gate_up_weight.size(1) // 128
               and gate_up_weight_scale.size(2) == (gate_up_weight.size(2) // 128 + 3) // 4 * 4,
               "gate_up_weight must be per 128 blockwise quant and must be aligned to 4")
    _C.require(down_weight_scale.size(1) == down_weight.size(1) // 128
               and down_weight_scale.size(2) == (down_weight.size(2) // 128 + 3) // 4 * 4,
               "down_weight must be per 128 blockwise quant and must be aligned to 4")
    _C.require(down_weight.size(1) == x.size(1) and down_weight.size(2) * 2 == gate_up_weight.size(1),
               "down_weight must be [num_expert, hidden, intermediate]")
    num_tokens, hidden = x.shape
    num_experts, inter2 = gate_up_weight.size(0), gate_up_weight.size(1)
    num_topk = topk_ids.size(1)
    _C.require(num_topk <= 128, "num_topk must less than or equal to 128")
    if shared_output is not None:
        _cuda_contig(shared_output, "shared_output")
        _C.require(shared_output.dtype == torch.bfloat16, "shared_output tensor dtype must be bfloat16")
        _C.require(tuple(shared_output.shape) == (num_tokens, hidden),
                   "shared_output tensor shape must be same as x tensor")
    if output is not None:
        _C.require(tuple(output.shape) == (num_tokens, hidden), "output shape must be [num_tokens, hidden_size]")
        _C.require(output.dtype == torch.bfloat16 and output.is_cuda, "output must be a cuda bfloat16 tensor")
        y = output
    else:
        y = torch.empty((num_tokens, hidden), dtype=torch.bfloat16, device=x.device)
    nbytes = _C.lib.hpc_fuse_moe_blockwise_workspace_bytes(num_tokens, num_topk, hidden, inter2, num_experts)
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=x.device)
    rc = _C.lib.hpc_fuse_moe_blockwise_async(
        _C.ptr(y), _C.ptr(ws), _C.ptr(x), _C.ptr(x_scale), _C.ptr(gate_up_weight),
        _C.ptr(gate_up_weight_scale), _C.ptr(down_weight), _C.ptr(down_weight_scale),
        _C.ptr(topk_ids), _C.ptr(topk_scale), _C.ptr(shared_output), num_tokens, hidden, inter2,
        num_topk, int(num_expert_total), num_experts, gate_up_weight_scale.size(2),
        down_weight_scale.size(2), int(rank_ep), _C.stream_of(x),
    )
    _C.check(rc, "fuse_moe_blockwise_async")
    return y


_T.impl("fuse_moe_blockwise_fp8", _fuse_moe_blockwise_entry, "CUDA")
_T.impl("fuse_moe_blockwise", _fuse_moe_blockwise_entry, "CUDA")


def _reduce_entry(x, topk_pos, topk_scale, shared_output):
    _cuda_contig(x, "x")
    _cuda_contig(topk_pos, "topk_pos")
    _cuda_contig(topk_scale, "topk_scale")
    _C.require(x.dtype == torch.bfloat16, "x dtype must be bfloat16")
    _C.require(topk_pos.dtype == torch.int32 and topk_scale.dtype == torch.float32,
               "topk_pos must be int32 and topk_scale float32")
    _C.require(topk_pos.shape == topk_scale.shape, "topk_pos and topk_scale must share the same shape")
    if shared_output is not None:
        _cuda_contig(shared_output, "shared_output")
        _C.require(shared_output.dtype == torch.bfloat16, "shared_output dtype must be bfloat16")
    num_tokens, num_topk = topk_pos.shape
    y = torch.empty((num_tokens, x.size(1)), dtype=torch.bfloat16, device=x.device)
    rc = _C.lib.hpc_moe_reduce_async(_C.ptr(y), _C.ptr(x), _C.ptr(topk_pos), _C.ptr(topk_scale),
                                     _C.ptr(shared_output), num_tokens, num_topk, x.size(1),
                                     _C.stream_of(x))
    _C.check(rc, "reduce_async")
    return y


_T.impl("reduce", _reduce_entry, "CUDA")


def _group_gemm_blockwise_fp8_entry(x, weight, seqlens, cu_seqlens, x_scale, w_scale,
                                    num_seq_per_group_avg, output, tma_desc, task_map_workspace):
    for t, n in ((x, "x"), (weight, "weight"), (seqlens, "seqlens"), (cu_seqlens, "cu_seqlens")):
        _C.require(t.is_cuda, f"{n} tensor must be cuda")
    _C.require(x.is_contiguous() and weight.is_contiguous(), "x / weight tensor must be contiguous")
    _C.require(x.dtype == _F8 and weight.dtype == _F8, "x and weight dtype must be fp8_e4m3")
    _C.require(seqlens.dtype == torch.int32 and cu_seqlens.dtype == torch.int32,
               "seqlens and cu_seqlens dtype must be int32")
    _C.require(x_scale.dtype == torch.float32 and w_scale.dtype == torch.float32,
               "x_scale and w_scale dtype must be float32")
    _C.require(x_scale.is_contiguous() and w_scale.is_contiguous(), "scales must be contiguous")
    _C.require(seqlens.size(0) == weight.size(0), "seqlens and weight must share the same num_group")
    _C.require(x.size(1) == weight.size(2), "x and weight must share the same k")
    _C.require(w_scale.size(2) % 4 == 0, "w_scale must be multiple of 4")
    m, k = x.shape
    n, num_group, m_pad = weight.size(1), seqlens.size(0), x_scale.size(1)
    y = output if output is not None else torch.empty((m, n), dtype=torch.bfloat16, device=x.device)
    tile_m = aligned_size(int(num_seq_per_group_avg))
    tiles = torch.empty(num_group, dtype=torch.int32, device=x.device)
    cu_tiles = torch.empty(num_group + 1, dtype=torch.int32, device=x.device)
    s = _C.stream_of(x)
    _C.check(_C.lib.hpc_moe_tiles_async(_C.ptr(seqlens), num_group, tile_m, _C.ptr(tiles),
                                        _C.ptr(cu_tiles), s), "group_gemm tiles")
    cu128 = _cu_tiles128(seqlens, m, num_group, s)
    rc = _C.lib.hpc_group_gemm_blockwise_fp8_async(
        _C.ptr(y), _C.ptr(x), _C.ptr(weight), _C.ptr(seqlens), _C.ptr(cu_seqlens), _C.ptr(x_scale),
        _C.ptr(w_scale), None, _C.ptr(cu_tiles), num_group, m, n, k, w_scale.size(2), tile_m, 1,
        m_pad, _C.ptr(cu128), s)
    _C.check(rc, "group_gemm_blockwise_fp8_async")
    del cu128  # lived until the launch was enqueued; stream order protects the rest
    return y


_T.impl("group_gemm_blockwise_fp8", _group_gemm_blockwise_fp8_entry, "CUDA")


# ---- per-tensor FP8 path (reference src/fuse_moe/entry.cc:18-443, src/group_gemm/entry.cc:14-89,
#      src/activation/entry.cc) ---------------------------------------------------------------------------
_T.define(
    "fuse_moe(Tensor x, Tensor gate_up_weight, Tensor down_weight, Tensor gate_up_scale, "
    "Tensor down_scale, Tensor act_and_mul_scale, Tensor topk_ids, Tensor topk_scale, Tensor ? "
    "shared_output, int rank_ep, int num_expert_total, bool use_bf16_mul, Tensor ? output) -> (Tensor)"
)
_T.define(
    "fuse_moe_pertensor_fp8(Tensor x, Tensor gate_up_weight, Tensor down_weight, Tensor "
    "gate_up_scale, Tensor down_scale, Tensor act_and_mul_scale, Tensor topk_ids, Tensor "
    "topk_scale, Tensor ? shared_output, int rank_ep, int num_expert_total, bool use_bf16_mul, "
    "Tensor ? output) -> (Tensor)"
)
_T.define(
    "count_and_gather(Tensor x, Tensor topk_ids, int num_expert, int rank_ep, int "
    "intermediate_size, int num_seq_per_group_avg) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, "
    "Tensor, Tensor, Tensor)"
)
_T.define(
    "group_gemm_fp8(Tensor x, Tensor weight, Tensor seqlens, Tensor cu_seqlens, Tensor y_scale, "
    "int num_seq_per_group_avg, Tensor? output, Tensor? tma_desc, Tensor? task_map_workspace) -> (Tensor)"
)
_T.define(
    "group_gemm_pertensor_fp8(Tensor x, Tensor weight, Tensor seqlens, Tensor cu_seqlens, Tensor y_scale, "
    "int num_seq_per_group_avg, Tensor? output, Tensor? tma_desc, Tensor? task_map_workspace) -> (Tensor)"
)
_T.define(  # verbatim: reference src/activation/entry.cc:208
    "act_mul_and_quant(Tensor input, Tensor scale, bool use_bf16_mul, Tensor? output) -> (Tensor)"
)
_T.define(  # verbatim: reference src/activation/entry.cc:223
    "scaled_fp8_quant(Tensor input, Tensor? scale, Tensor? output) -> (Tensor, Tensor)"
)


def _fuse_moe_entry(x, gate_up_weight, down_weight, gate_up_scale, down_scale, act_and_mul_scale,
                    topk_ids, topk_scale, shared_output, rank_ep, num_expert_total, use_bf16_mul, output):
    _C.require(x.dtype == _F8 and gate_up_weight.dtype == _F8 and down_weight.dtype == _F8,
               "x, gate_up_weight and down_weight dtype must be fp8_e4m3")
    _C.require(topk_ids.dtype == torch.int32, "topk_ids dtype must be int32")
    _C.require(gate_up_scale.dtype == torch.float32 and down_scale.dtype == torch.float32
               and act_and_mul_scale.dtype == torch.float32 and topk_scale.dtype == torch.float32,
               "gate_up_scale, down_scale, act_and_mul_scale and topk_scale dtype must be float32")
    for t, n in ((x, "x"), (gate_up_weight, "gate_up_weight"), (gate_up_scale, "gate_up_scale"),
                 (down_weight, "down_weight"), (down_scale, "down_scale"), (topk_ids, "topk_ids"),
                 (topk_scale, "topk_scale"), (act_and_mul_scale, "act_and_mul_scale")):
        _cuda_contig(t, n)
    _C.require(x.size(0) == topk_ids.size(0), "x and topk_ids must share the same num_seq")
    _C.require(topk_ids.shape == topk_scale.shape, "topk_ids and topk_scale must share the same shape")
    _C.require(x.size(1) == gate_up_weight.size(2), "x and weight must share the same k")
    _C.require(gate_up_weight.size(0) == down_weight.size(0),
               "gate_up_weight and down_weight must share the same num_expert")
    num_tokens, hidden = x.shape
    num_experts, inter2 = gate_up_weight.size(0), gate_up_weight.size(1)
    num_topk = topk_ids.size(1)
    _C.require(num_topk <= 128, "num_topk must less than or equal to 128")
    _C.require(gate_up_scale.numel() >= num_experts and down_scale.numel() >= num_experts,
               "one scale per local expert is required")
    if shared_output is not None:
        _cuda_contig(shared_output, "shared_output")
        _C.require(shared_output.dtype == torch.bfloat16, "shared_output tensor dtype must be bfloat16")
        _C.require(tuple(shared_output.shape) == (num_tokens, hidden),
                   "shared_output tensor shape must be same as x tensor")
    if output is not None:
        _C.require(tuple(output.shape) == (num_tokens, hidden) and output.dtype == torch.bfloat16
                   and output.is_cuda, "output must be a cuda bfloat16 [num_tokens, hidden_size] tensor")
        y = output
    else:
        y = torch.empty((num_tokens, hidden), dtype=torch.bfloat16, device=x.device)
    nbytes = _C.lib.hpc_fuse_moe_blockwise_workspace_bytes(num_tokens, num_topk, hidden,
                                                           (inter2 + 255) // 256 * 256, num_experts)
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=x.device)
    rc = _C.lib.hpc_fuse_moe_pertensor_async(
        _C.ptr(y), _C.ptr(ws), _C.ptr(x), _C.ptr(gate_up_weight), _C.ptr(down_weight),
        _C.ptr(gate_up_scale), _C.ptr(down_scale), _C.ptr(act_and_mul_scale), _C.ptr(topk_ids),
        _C.ptr(topk_scale), _C.ptr(shared_output), num_tokens, hidden, inter2, num_topk, num_experts,
        int(rank_ep), int(bool(use_bf16_mul)), _C.stream_of(x))
    _C.check(rc, "fuse_moe_async")
    return y


_T.impl("fuse_moe", _fuse_moe_entry, "CUDA")
_T.impl("fuse_moe_pertensor_fp8", _fuse_moe_entry, "CUDA")


def _count_and_gather_entry(x, topk_ids, num_expert, rank_ep, intermediate_size, num_seq_per_group_avg):
    _cuda_contig(x, "x")
    _cuda_contig(topk_ids, "topk_ids")
    _C.require(x.size(0) == topk_ids.size(0), "x and topk_ids must share the same k")
    _C.require(topk_ids.dtype == torch.int32 and x.element_size() == 1, "x must be fp8, topk_ids int32")
    num_seq, hidden = x.shape
    num_topk = topk_ids.size(1)
    dev = x.device
    i32 = dict(dtype=torch.int32, device=dev)
    gate_up_input = torch.empty((num_seq * num_topk, hidden), dtype=x.dtype, device=dev)
    gate_up_output = torch.empty((num_seq * num_topk, intermediate_size), dtype=torch.bfloat16, device=dev)
    topk_pos = torch.empty((num_seq, num_topk), **i32)
    seqlens = torch.zeros(num_expert, **i32)
    cu_seqlens = torch.empty(num_expert + 1, **i32)
    tiles = torch.empty(num_expert, **i32)
    cu_tiles = torch.empty(num_expert + 1, **i32)
    row_index = torch.empty(num_seq * num_topk, **i32)
    tmas = torch.empty((num_expert * 2, 128), dtype=torch.int8, device=dev)  # unused: no TMA on gfx950
    tile_m = aligned_size(int(num_seq_per_group_avg))
    s = _C.stream_of(x)
    _C.check(_C.lib.hpc_moe_count_and_slot_async(
        _C.ptr(topk_ids), num_seq, num_topk, int(num_expert), int(rank_ep), tile_m, _C.ptr(seqlens),
        _C.ptr(cu_seqlens), _C.ptr(tiles), _C.ptr(cu_tiles), _C.ptr(topk_pos), _C.ptr(row_index), s),
        "count_and_gather_async")
    _C.check(_C.lib.hpc_moe_gather_rows_async(_C.ptr(x), _C.ptr(topk_pos), num_seq, num_topk, hidden,
                                              _C.ptr(gate_up_input), s), "count_and_gather_async")
    return (gate_up_input, gate_up_output, topk_pos, seqlens, cu_seqlens, tiles, cu_tiles, tmas, tmas.clone())


_T.impl("count_and_gather", _count_and_gather_entry, "CUDA")


def _group_gemm_fp8_entry(x, weight, seqlens, cu_seqlens, y_scale, num_seq_per_group_avg, output,
                          tma_desc, task_map_workspace):
    for t, n in ((x, "x"), (weight, "weight"), (seqlens, "seqlens"), (cu_seqlens, "cu_seqlens"),
                 (y_scale, "y_scale")):
        _C.require(t.is_cuda, f"{n} tensor must be cuda")
    _C.require(x.is_contiguous() and weight.is_contiguous(), "x / weight tensor must be contiguous")
    _C.require(x.dtype == _F8 and weight.dtype == _F8, "x and weight dtype must be fp8_e4m3")
    _C.require(seqlens.dtype == torch.int32 and cu_seqlens.dtype == torch.int32,
               "seqlens and cu_seqlens dtype must be int32")
    _C.require(y_scale.dtype == torch.float32 and y_scale.numel() >= weight.size(0),
               "y_scale must be float32 [num_group]")
    _C.require(seqlens.size(0) == weight.size(0), "seqlens and weight must share the same num_group")
    _C.require(x.size(1) == weight.size(2), "x and weight must share the same k")
    m, k = x.shape
    n, num_group = weight.size(1), seqlens.size(0)
    y = output if output is not None else torch.empty((m, n), dtype=torch.bfloat16, device=x.device)
    s = _C.stream_of(x)
    cu128 = _cu_tiles128(seqlens, m, num_group, s)
    rc = _C.lib.hpc_group_gemm_pertensor_fp8_async(
        _C.ptr(y), _C.ptr(x), _C.ptr(weight), _C.ptr(seqlens), _C.ptr(cu_seqlens), _C.ptr(y_scale), None,
        num_group, m, m, n, k, _C.ptr(cu128), s)
    _C.check(rc, "group_gemm_fp8_async")
    del cu128
    return y


_T.impl("group_gemm_fp8", _group_gemm_fp8_entry, "CUDA")
_T.impl("group_gemm_pertensor_fp8", _group_gemm_fp8_entry, "CUDA")


def _act_mul_and_quant_entry(gate_up, scale, use_bf16_mul, output):
    _cuda_contig(gate_up, "gate_up")
    _C.require(gate_up.dtype == torch.bfloat16 and gate_up.dim() == 2, "gate_up must be bfloat16 [N, 2*C]")
    _C.require(scale.is_cuda and scale.dtype == torch.float32, "scale must be a cuda float32 tensor")
    rows, inter = gate_up.size(0), gate_up.size(1) // 2
    out = output if output is not None else torch.empty((rows, inter), dtype=_F8, device=gate_up.device)
    rc = _C.lib.hpc_act_mul_and_quant_async(_C.ptr(out), _C.ptr(gate_up), _C.ptr(scale), None, rows, inter,
                                            int(bool(use_bf16_mul)), _C.stream_of(gate_up))
    _C.check(rc, "act_mul_and_quant_async")
    return out


_T.impl("act_mul_and_quant", _act_mul_and_quant_entry, "CUDA")


_QUANT_IN = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}


def _scaled_fp8_quant_entry(input, scale, output):
    """reference scaled_fp8_quant_entry, src/activation/entry.cc:158-200: out = e4m3(input * (1 / scale[0])),
    returns (output, scale)."""
    _C.require(input.is_cuda, "input must be a CUDA tensor")
    _C.require(input.is_contiguous(), "input must be contiguous")
    _C.require(input.numel() > 0, "input must be non-empty")
    _C.require(input.dtype in _QUANT_IN, "input dtype must be float32, float16, or bfloat16")
    out = output if output is not None else torch.empty_like(input, dtype=_F8)
    _C.require(out.is_cuda, "output must be a CUDA tensor")
    _C.require(out.is_contiguous(), "output must be contiguous")
    _C.require(out.shape == input.shape, "output shape must match input shape")
    _C.require(out.dtype == _F8, "output dtype must be float8_e4m3fn")
    _C.require(scale is not None, "scale is required for scaled_fp8_quant")
    _C.require(scale.is_cuda, "scale must be a CUDA tensor")
    _C.require(scale.dtype == torch.float32, "scale dtype must be float32")
    _C.require(scale.numel() == 1, "scale must contain one element")
    rc = _C.lib.hpc_scaled_fp8_quant_async(_C.ptr(out), _C.ptr(input), _C.ptr(scale), input.numel(),
                                           _QUANT_IN[input.dtype], _C.stream_of(input))
    _C.check(rc, "scaled_fp8_quant_async")
    return out, scale


_T.impl("scaled_fp8_quant", _scaled_fp8_quant_entry, "CUDA")


# ---- DeepEP-format helper + the reference's raw cp.async group-GEMM ops ------------------------------------
_T.define("reformat_x_scale(Tensor x_scale, Tensor seqlens, Tensor cu_seqlens, "
          "Tensor? out_x_scale, int num_seq_per_group_avg) -> (Tensor)")
_T.define("group_gemm_fp8_cp_async(Tensor x, Tensor weight, Tensor y_scale, Tensor seqlens, Tensor "
          "cu_seqlens, Tensor tiles, Tensor cu_tiles, bool use_task_map=False) -> (Tensor)")
_T.define("group_gemm_fp8_scatter_cp_async(Tensor x, Tensor weight, Tensor y_scale, Tensor "
          "row_indices, Tensor seqlens, Tensor cu_seqlens, Tensor tiles, "
          "Tensor cu_tiles, bool use_task_map=False) -> (Tensor)")


def _reformat_x_scale_entry(x_scale, seqlens, cu_seqlens, out_x_scale, num_seq_per_group_avg):
    # reference reformat_x_scale_entry, src/group_gemm/entry.cc:170-222
    for t, name in ((x_scale, "x_scale"), (seqlens, "seqlens"), (cu_seqlens, "cu_seqlens")):
        _C.require(t.is_cuda, f"{name} tensor must be cuda")
        _C.require(t.is_contiguous(), f"{name} tensor a must be contiguous")
    _C.require(x_scale.dtype == torch.float32 and x_scale.dim() == 2, "x_scale must be float32 [rows, K/128]")
    m, n = x_scale.shape
    num_group = seqlens.size(0)
    avg = int(num_seq_per_group_avg)
    tilem = 8 if avg <= 8 else 16 if avg <= 16 else 32 if avg <= 32 else 48 if avg <= 48 else 64
    _C.require((m // num_group) % tilem == 0,
               "The sparse pad length of x_scale for each group must be aligned to multiple of "
               "8/16/32/48/64 according to num_seq_per_group_avg")
    out = out_x_scale if out_x_scale is not None else torch.empty((n, m), dtype=x_scale.dtype, device=x_scale.device)
    _C.require(out.is_contiguous() and out.dtype == torch.float32 and out.numel() >= n * m,
               "out_x_scale must be a contiguous float32 [K/128, rows] tensor")
    _C.check(_C.lib.hpc_reformat_x_scale_async(_C.ptr(out), _C.ptr(x_scale), _C.ptr(seqlens), _C.ptr(cu_seqlens),
                                               num_group, m, n, tilem, _C.stream_of(x_scale)), "reformat_x_scale")
    return out


def _group_gemm_cp_async(x, weight, y_scale, row_indices, seqlens, cu_seqlens):
    _cuda_contig(x, "x")
    _cuda_contig(weight, "weight")
    _C.require(x.dtype == torch.float8_e4m3fn and weight.dtype == torch.float8_e4m3fn, "x / weight must be fp8_e4m3")
    _C.require(y_scale.dtype == torch.float32 and seqlens.dtype == torch.int32 and cu_seqlens.dtype == torch.int32,
               "y_scale must be float32, seqlens / cu_seqlens int32")
    num_group, n, k = weight.shape
    m = row_indices.size(0) if row_indices is not None else x.size(0)
    y = torch.empty((m, n), dtype=torch.bfloat16, device=x.device)
    s = _C.stream_of(x)
    cu128 = _cu_tiles128(seqlens, m, num_group, s)
    rc = _C.lib.hpc_group_gemm_pertensor_fp8_async(
        _C.ptr(y), _C.ptr(x), _C.ptr(weight), _C.ptr(seqlens), _C.ptr(cu_seqlens), _C.ptr(y_scale),
        _C.ptr(row_indices), num_group, m, x.size(0), n, k, _C.ptr(cu128), s)
    _C.check(rc, "group_gemm_fp8_cp_async")
    del cu128
    return y


def _group_gemm_fp8_cp_async_entry(x, weight, y_scale, seqlens, cu_seqlens, tiles, cu_tiles, use_task_map=False):
    # reference src/group_gemm/cp_async/entry.cc (the caller's 64-row tile tables are not needed here)
    return _group_gemm_cp_async(x, weight, y_scale, None, seqlens, cu_seqlens)


def _group_gemm_fp8_scatter_cp_async_entry(x, weight, y_scale, row_indices, seqlens, cu_seqlens, tiles, cu_tiles,
                                           use_task_map=False):
    _C.require(row_indices.is_cuda and row_indices.dtype == torch.int32 and row_indices.is_contiguous(),
               "row_indices must be a contiguous cuda int32 tensor")
    return _group_gemm_cp_async(x, weight, y_scale, row_indices, seqlens, cu_seqlens)


_T.impl("reformat_x_scale", _reformat_x_scale_entry, "CUDA")
_T.impl("group_gemm_fp8_cp_async", _group_gemm_fp8_cp_async_entry, "CUDA")
_T.impl("group_gemm_fp8_scatter_cp_async", _group_gemm_fp8_scatter_cp_async_entry, "CUDA")


# ---- masked (DeepEP-layout) activation variants (reference src/activation/entry.cc:50-110) ---------------------
_T.define("masked_act_mul_and_quant(Tensor input, Tensor scale, Tensor num_per_expert, Tensor? output) -> (Tensor)")
_T.define(  # verbatim: reference src/activation/entry.cc:217
    "masked_act_mul_and_blockwise_quant(Tensor input, Tensor num_per_expert, Tensor? output, Tensor? "
    "output_scale) -> (Tensor output, Tensor output_scale)"
)


def _masked_common(input, num_per_expert):
    _C.require(input.is_contiguous(), "input tensor must be contiguous")
    _C.require(num_per_expert.is_contiguous(), "num_per_expert tensor must be contiguous")
    _C.require(input.is_cuda, "input tensor's device must be cuda")
    _C.require(num_per_expert.is_cuda, "num_per_expert tensor's device must be cuda")
    _C.require(input.dtype == torch.bfloat16 and input.dim() == 2, "input must be bfloat16 [N, 2*C]")
    _C.require(num_per_expert.dtype == torch.int32, "num_per_expert must be int32")
    num_experts, total = num_per_expert.size(0), input.size(0)
    _C.require(num_experts > 0 and total % num_experts == 0, "rows must be num_expert * num_token_padded_per_expert")
    return total, input.size(1) // 2, total // num_experts


def _masked_act_mul_and_quant_entry(input, scale, num_per_expert, output=None):
    total, inter, per = _masked_common(input, num_per_expert)
    _C.require(scale.is_contiguous() and scale.is_cuda, "scale tensor must be contiguous, on cuda")
    _C.require(inter % 8 == 0, "hidden dim must be divided by 8")
    _C.require(scale.numel() == 1 and scale.dtype == torch.float32, "only support per tensor qunat")
    out = output if output is not None else torch.empty((total, inter), dtype=_F8, device=input.device)
    _C.check(_C.lib.hpc_masked_act_mul_and_quant_async(_C.ptr(out), _C.ptr(input), _C.ptr(scale), _C.ptr(num_per_expert),
                                                       total, inter, per, _C.stream_of(input)),
             "masked_act_mul_and_quant_async")
    return out


def _masked_act_mul_and_blockwise_quant_entry(input, num_per_expert, output=None, output_scale=None):
    total, inter, per = _masked_common(input, num_per_expert)
    _C.require(inter % 128 == 0, "hidden dim must be divided by 128")
    out = output if output is not None else torch.empty((total, inter), dtype=_F8, device=input.device)
    osc = output_scale if output_scale is not None else torch.empty((total, inter // 128), dtype=torch.float32,
                                                                    device=input.device)
    _C.require(osc.is_contiguous() and osc.dtype == torch.float32, "output_scale must be contiguous float32")
    _C.check(_C.lib.hpc_masked_act_mul_and_blockwise_quant_async(_C.ptr(out), _C.ptr(osc), _C.ptr(input),
                                                                 _C.ptr(num_per_expert), total, inter, per,
                                                                 _C.stream_of(input)),
             "masked_act_mul_and_blockwise_quant_async")
    return out, osc


_T.impl("masked_act_mul_and_quant", _masked_act_mul_and_quant_entry, "CUDA")
_T.impl("masked_act_mul_and_blockwise_quant", _masked_act_mul_and_blockwise_quant_entry, "CUDA")
