"""hpc.attention — decode-attention surface of the reference (hpc/attention.py:8-12, :336-696).

Same function names, argument names/defaults and semantics; the kernels behind torch.ops.hpc.* are
the gfx950 ones of libhpc_amd.so.  Prefill / block-sparse entry points of the reference are out of
scope of this hot path and intentionally absent.
"""
from enum import Enum

import torch
from torch import Tensor

from . import _C


class QuantType(Enum):
    QPERTOKEN_PERHEAD_KPERTOKEN_PERHEAD_VPERHEAD = 0
    QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR = 1
    QPERTENSOR_KPERTENSOR_VPERTENSOR = 2
    QPERTOKEN_PERHEAD_KPERTOKEN_PERHEAD_VPERHEAD_QKHADAMARD = 3


def attention_decode_bf16(
    q: Tensor,
    kcache: Tensor,
    vcache: Tensor,
    block_ids: Tensor,
    num_seq_kvcache: Tensor,
    mtp: int = 0,
    new_kv_included: bool = False,
    splitk: bool = True,
    task_map: Tensor = None,
    split_flag: Tensor = None,
    output: Tensor = None,
) -> Tensor:
    """Paged decode attention in bfloat16 (reference hpc/attention.py:336-417).

    Args:
        q: [num_batch * num_seq_q, num_head_q, 128] bfloat16 (num_seq_q = mtp + 1).
        kcache / vcache: paged caches, logical [num_blocks, block_size, num_head_kv, 128] bfloat16;
            any block/token/head strides (NHD-contiguous and HND-backed views both work).  Unused
            slots of a request's last block should be zero (they are masked anyway).
        block_ids: [num_batch, max_blocks] int32 page table.
        num_seq_kvcache: [num_batch] int32; tokens in the cache before this step, or including the
            num_seq_q new ones when new_kv_included.
        splitk: accepted for API compatibility; the schedule is always the dynamic tile schedule.
        task_map: workspace from get_attention_decode_task_workspace filled by
            assign_attention_decode_task (scheduled on the fly when None).
        split_flag: unused on MI355X (the reference's static path spins on it).
        output: optional preallocated [num_batch * num_seq_q, num_head_q, 128] bfloat16.
    """
    return torch.ops.hpc.attention_decode_bf16(
        q, kcache, vcache, block_ids, num_seq_kvcache, mtp, new_kv_included, splitk, task_map,
        split_flag, output,
    )


def attention_decode_fp8(
    q: Tensor,
    kcache: Tensor,
    vcache: Tensor,
    block_ids: Tensor,
    num_seq_kvcache: Tensor,
    qscale: Tensor,
    kscale: Tensor,
    vscale: Tensor,
    mtp: int = 0,
    new_kv_included: bool = False,
    quant_type: QuantType = QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR,
    splitk: bool = True,
    task_map: Tensor = None,
    split_flag: Tensor = None,
    output: Tensor = None,
) -> Tensor:
    """Paged decode attention with FP8 (e4m3) Q/K/V (reference hpc/attention.py:420-517):
    softmax(Q K^T * qscale * kscale / sqrt(head_dim)) V * vscale, bfloat16 output.

    Args (beyond attention_decode_bf16):
        q: [num_batch * num_seq_q, num_head_q, 128] float8_e4m3fn.
        kcache / vcache: paged caches of 1-byte elements (float8_e4m3fn).
        qscale: float32 [num_batch * num_seq_q, num_head_q] per-token per-head Q scale.
        kscale: float32 [1] (QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR) or the view of the K-cache
            tail rows `kcache_full[:, block_size:]` holding per-token per-head fp32 scales
            (QPERTOKEN_PERHEAD_KPERTOKEN_PERHEAD_VPERHEAD).
        vscale: float32 [1] or [num_head_kv].
    """
    return torch.ops.hpc.attention_decode_fp8(
        q, kcache, vcache, block_ids, num_seq_kvcache, qscale, kscale, vscale, mtp,
        new_kv_included, quant_type.value, splitk, task_map, split_flag, output,
    )


def get_attention_decode_task_workspace(
    max_num_batch: int, max_seqlen: int, num_head_kv: int, min_process_len: int = 512
):
    """Allocate the task-map workspace (reference hpc/attention.py:520-582; same byte layout).

    Returns int8 [task_map_byte_size] on the current device with header ints 2..4 pre-filled
    (num_head_kv, max_num_batch, scheduler byte size).
    """
    dev = torch.device("cuda", torch.cuda.current_device())
    num_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    total, sched = task_workspace_bytes(num_cu, max_num_batch, max_seqlen, num_head_kv, min_process_len)
    ws = torch.zeros(total, dtype=torch.int8, device=dev)
    hdr = ws.view(torch.int32)
    hdr[2] = num_head_kv
    hdr[3] = max_num_batch
    hdr[4] = sched
    return ws


def task_workspace_bytes(num_cu, max_num_batch, max_seqlen, num_head_kv, min_process_len):
    """Byte size + scheduler byte size of the task-map workspace; same arithmetic as reference
    hpc/attention.py:540-571 (sized for up to 4 bins per CU, so any gfx950 bin count fits)."""
    k_task, k_max_cta, k_tile = 48, 4, 64
    max_cta = num_cu * k_max_cta
    total_tiles = max_num_batch * num_head_kv * ((max_seqlen + k_tile - 1) // k_tile)
    max_tasks = 0
    for cta_per_cu in (4, 3, 2, 1):
        ctas = num_cu * cta_per_cu
        per = max((total_tiles + ctas - 1) // ctas, min_process_len // k_tile)
        max_tasks = max(max_tasks, (per + 1) * ctas + 1)
    chunk_bytes = (max_num_batch * num_head_kv * 4 + k_task - 1) // k_task * k_task
    cta_pad = (max_cta + 11) // 12 * 12 * 4
    sched = max_tasks * k_task + chunk_bytes
    return sched + 2 * cta_pad, sched


def release_decode_workspaces():
    """Drop EVERY zero-once scratch buffer the host library caches per (purpose, device, stream, hipGraph capture) - the
    decode workspaces and the router GEMM's split-K ticket buffers alike (csrc/torch_common.h::release_cached_scratch) -
    e.g. after the streams / graphs that used them are gone.  Buffers of captures that have ended are dropped on the
    next call on their stream anyway."""
    torch.ops.hpc._release_decode_workspaces()


def assign_attention_decode_task(
    num_seq_kvcache: Tensor,
    task_map: Tensor,
    num_head_kv: int,
    mtp: int,
    new_kv_included: bool,
    min_process_len: int = 512,
) -> Tensor:
    """Fill `task_map` for one decode step (reference hpc/attention.py:585-626).

    NOTE: as in the reference, the 4th argument is passed to the op as num_seq_q (callers pass
    num_seq_q = mtp + 1 here, tests/test_attention_decode_bf16.py:105-121 of the reference).
    `num_seq_kvcache` may live on the GPU (device scheduler) or on the CPU (host scheduler, then
    three byte ranges are copied into the device workspace); both give byte-identical maps.

    `min_process_len` (KV tokens one workgroup processes at least) shapes the map's bins for the kernels that consume
    the map (<= 16 q rows per kv head on HND pages or with an odd kv-head count).  The head-pair kernels (NHD pages, even
    kv-head count; 17-32 q rows per kv head on any layout: csrc/attention_decode_v2.hip) plan their ranges in closed form inside the launch - the map's bins and split
    decisions do not apply there - and honour this lower bound through header int 6 of the map, where the scheduler
    records it (csrc/sched_task_info.h)."""
    if num_seq_kvcache.device.type == "cpu":
        task_map_host = torch.ops.hpc.assign_attention_decode_task(
            num_seq_kvcache, num_head_kv, mtp, new_kv_included, min_process_len, None
        )
        flat = task_map_host.reshape(-1)
        task_map[:8].copy_(flat[:8], non_blocking=True)
        task_map[20:28].copy_(flat[20:28], non_blocking=True)  # max chunks per request, min_process_len (header int 6)
        task_map[48 : flat.numel()].copy_(flat[48:], non_blocking=True)
        return task_map
    return torch.ops.hpc.assign_attention_decode_task(
        num_seq_kvcache, num_head_kv, mtp, new_kv_included, min_process_len, task_map
    )


def print_attention_decode_task(task_map: Tensor) -> None:
    """Pretty-print a task map (reference hpc/attention.py:629-696)."""
    stride = 12
    task = task_map.view(torch.int32).reshape(-1)
    task = task[: task.numel() // stride * stride].reshape(-1, stride).cpu()
    per1, bins = int(task[0][0]), int(task[0][1])
    num_head_kv, max_num_batch = int(task[0][2]), int(task[0][3])
    chunk_row = 1 + bins * per1
    chunks = task[chunk_row:].reshape(-1)[: num_head_kv * max_num_batch]
    print(
        f"\n[Dynamic Decode Attn Task Map] num_tile_per_cta={per1 - 1}, num_head_kv={num_head_kv}, "
        f"max_num_batch={max_num_batch}, num_total_ctas={bins}"
    )
    print(f"num_chunks[ihead_kv, ibatch]:\n{chunks.reshape(num_head_kv, max_num_batch)}\n")
    idx, empty = 0, 0
    for icta in range(bins):
        rows = task[1 + icta * per1 : 1 + (icta + 1) * per1]
        if int(rows[0][0]) < 0 or int(rows[0][1]) < 0:
            empty += 1
            continue
        print(f"#######CTA{icta}########")
        ntask, seqkv = 0, 0
        for row in rows[: per1 - 1]:
            r = [int(v) for v in row]
            if r[0] < 0 or r[1] < 0:
                break
            print(
                f"task:{idx}, ihead_kv:{r[0]}, ibatch:{r[1]}, ichunk:{r[2]}, iseq_start:{r[3]}, "
                f"num_seqkv:{r[4]}, num_seqkvcache:{r[5]}, num_tile_kv:{r[6]}, "
                f"num_tile_full:{r[7]}, is_casual_chunk:{r[8]}"
            )
            idx += 1
            ntask += 1
            seqkv += r[4]
        print(f"CTA:{icta}, num_tasks:{ntask}, total_seqkv:{seqkv}")
    print(f"[idle] {empty}/{bins} bins were empty")


@torch.library.register_fake("hpc::attention_decode_bf16")
def attention_decode_bf16_fake(
    q, kcache, vcache, block_ids, num_seq_kvcache, mtp, new_kv_included, splitk,
    task_map=None, split_flag=None, output=None,
):
    return torch.empty_like(q)


@torch.library.register_fake("hpc::attention_decode_fp8")
def attention_decode_fp8_fake(
    q, kcache, vcache, block_ids, num_seq_kvcache, qscale, kscale, vscale, mtp, new_kv_included,
    quant_type, splitk, task_map=None, split_flag=None, output=None,
):
    return torch.empty_like(q, dtype=torch.bfloat16)


def attention_with_kvcache_prefill_fp8(
    q: Tensor,
    kcache: Tensor,
    vcache: Tensor,
    qscale: Tensor,
    kscale: Tensor,
    vscale: Tensor,
    cu_seqlens_q: Tensor,
    block_ids: Tensor,
    seqlens_kvcache: Tensor,
    max_seqlens_q: int,
    quant_type: QuantType = QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR,
    output: Tensor = None,
) -> Tensor:
    """Causal paged-KV prefill attention with FP8 q/k/v (reference hpc/attention.py:148-251).

    q e4m3 [total_seq, Hq, 128]; kcache / vcache e4m3 logical [num_blocks, block_size, Hkv, 128] (NHD or
    HND-backed strides); qscale f32 [num_batch, Hq, max_seqlens_q_pad] (per token, per head);
    kscale f32 [1] / vscale f32 [1] (per tensor) or the K-scale tail rows + vscale [Hkv];
    cu_seqlens_q int32 [num_batch+1]; block_ids int32 [num_batch, max_blocks]; seqlens_kvcache int32
    [num_batch]: tokens of each request in the cache, the request's q tokens being the last ones (q row s
    attends keys j <= L - Sq + s, as in the reference tests' oracle).  Returns bf16 [total_seq, Hq, 128]."""
    return torch.ops.hpc.attention_with_kvcache_prefill_fp8(
        q, kcache, vcache, qscale, kscale, vscale, cu_seqlens_q, block_ids, seqlens_kvcache, int(max_seqlens_q),
        quant_type.value, output)


@torch.library.register_fake("hpc::attention_with_kvcache_prefill_fp8")
def _attention_with_kvcache_prefill_fp8_fake(q, kcache, vcache, qscale, kscale, vscale, cu_seqlens_q, block_ids,
                                             seqlens_kvcache, max_seqlens_q, quant_type, output=None):
    if output is not None:
        return output
    return torch.empty((q.shape[0], q.shape[1], vcache.shape[-1]), dtype=torch.bfloat16, device=q.device)


def attention_with_kvcache_blocksparse_prefill_fp8(
    q: Tensor,
    kcache: Tensor,
    vcache: Tensor,
    qscale: Tensor,
    kscale: Tensor,
    vscale: Tensor,
    cu_seqlens_q: Tensor,
    block_ids: Tensor,
    seqlens_kvcache: Tensor,
    max_seqlens_q: int,
    quant_type: QuantType = QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR,
    block_mask: Tensor = None,
    output: Tensor = None,
) -> Tensor:
    """Dense / block-sparse causal prefill over the paged FP8 cache (reference hpc/attention.py:253-339).
    Arguments as attention_with_kvcache_prefill_fp8 plus block_mask uint8
    [num_batch, num_head_q, ceil(max_seqlens_q/128), Kb]: only 128 x 128 (q positions x kv tokens) tiles marked
    non-zero are attended (None = dense).  Keep the causal diagonal tile of every q tile set: a q row with no
    active key yields zeros here (NaN in the reference's PyTorch model)."""
    return torch.ops.hpc.attention_with_kvcache_blocksparse_prefill_fp8(
        q, kcache, vcache, qscale, kscale, vscale, cu_seqlens_q, block_ids, seqlens_kvcache, int(max_seqlens_q),
        quant_type.value, block_mask, output)


@torch.library.register_fake("hpc::attention_with_kvcache_blocksparse_prefill_fp8")
def _attention_blocksparse_prefill_fp8_fake(q, kcache, vcache, qscale, kscale, vscale, cu_seqlens_q, block_ids,
                                            seqlens_kvcache, max_seqlens_q, quant_type, block_mask=None, output=None):
    if output is not None:
        return output
    return torch.empty((q.shape[0], q.shape[1], vcache.shape[-1]), dtype=torch.bfloat16, device=q.device)


def attention_prefill_bf16(q: Tensor, k: Tensor, v: Tensor, seqlens_q: Tensor, cu_seqlens_q: Tensor,
                           max_seqlens_q: int, output: Tensor = None) -> Tensor:
    """Causal varlen prefill attention in bf16 over contiguous K/V (reference hpc/attention.py:15-68).
    q [total_seq, Hq, 128], k / v [total_seq, Hkv, 128] bf16; seqlens_q int32 [B]; cu_seqlens_q int32 [B+1];
    token s of a request attends tokens 0..s of the same request.  Returns bf16 [total_seq, Hq, 128]."""
    return torch.ops.hpc.attention_prefill_bf16(q, k, v, seqlens_q, cu_seqlens_q, int(max_seqlens_q), output)


def attention_with_kvcache_prefill_bf16(q: Tensor, kcache: Tensor, vcache: Tensor, cu_seqlens_q: Tensor,
                                        block_ids: Tensor, seqlens_kvcache: Tensor, max_seqlens_q: int,
                                        output: Tensor = None) -> Tensor:
    """Causal prefill attention in bf16 over the paged KV cache (reference hpc/attention.py:70-146).
    q [total_seq, Hq, 128]; kcache / vcache logical [num_blocks, block_size, Hkv, 128] (NHD or HND-backed
    strides); cu_seqlens_q int32 [B+1]; block_ids int32 [B, max_blocks]; seqlens_kvcache int32 [B] = cached
    tokens of each request, its q tokens being the last ones (q row s attends keys j <= L - Sq + s).
    Returns bf16 [total_seq, Hq, 128]."""
    return torch.ops.hpc.attention_with_kvcache_prefill_bf16(q, kcache, vcache, cu_seqlens_q, block_ids,
                                                             seqlens_kvcache, int(max_seqlens_q), output)


@torch.library.register_fake("hpc::attention_prefill_bf16")
def _attention_prefill_bf16_fake(q, k, v, seqlens_q, cu_seqlens_q, max_seqlens_q, output=None):
    return output if output is not None else torch.empty((q.shape[0], q.shape[1], v.shape[-1]), dtype=q.dtype,
                                                         device=q.device)


@torch.library.register_fake("hpc::attention_with_kvcache_prefill_bf16")
def _attention_with_kvcache_prefill_bf16_fake(q, kcache, vcache, cu_seqlens_q, block_ids, num_seq_kvcache,
                                              max_seqlens_q, output=None):
    return output if output is not None else torch.empty((q.shape[0], q.shape[1], vcache.shape[-1]), dtype=q.dtype,
                                                         device=q.device)
