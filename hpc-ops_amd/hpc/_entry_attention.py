"""torch.ops.hpc.{assign_attention_decode_task, attention_decode_bf16, attention_decode_fp8}.

Mirror of the decode part of reference src/attention/entry.cc (:411-817: checks, scratch
allocation, stride extraction, schema block :851-873); compute is in libhpc_amd.so
(csrc/assign_task.hip, csrc/attention_decode_*.hip) behind the C-ABI of include/hpc_amd.h.
"""
import ctypes

from ctypes import c_void_p

import torch

from . import _C

_T = _C.torch_lib

_T.define(
    "assign_attention_decode_task(Tensor num_seq_kvcache, int num_head_kv, int num_seq_q, bool "
    "new_kv_included, int min_process_len, Tensor? task_map) -> (Tensor)"
)
_T.define(
    "attention_decode_bf16(Tensor q, Tensor! kcache, Tensor! vcache, Tensor block_ids, Tensor "
    "num_seq_kvcache, int mtp, bool new_kv_included, bool use_splitk, Tensor? task_map, "
    "Tensor? split_flag, Tensor? output) -> (Tensor)"
)

_T.define(
    "attention_decode_fp8(Tensor q, Tensor! kcache, Tensor! vcache, Tensor block_ids, Tensor "
    "num_seq_kvcache, Tensor qscale, Tensor kscale, Tensor vscale, int mtp, bool "
    "new_kv_included, int quant_type, bool use_splitk, Tensor? task_map, Tensor? split_flag, "
    "Tensor? output) -> (Tensor)"
)

_INT_P = ctypes.POINTER(ctypes.c_int)


def num_bins(num_seq_q: int, device=None) -> int:
    """Scheduler bins (= decode workgroups) on this device; reference kCtaPerSmMap * SM count
    (src/attention/entry.cc:745-746), here CUs * workgroups-per-CU chosen for gfx950."""
    idx = -1
    if device is not None and getattr(device, "index", None) is not None:
        idx = device.index
    n = _C.lib.hpc_attention_decode_num_bins(int(num_seq_q), idx)
    _C.require(n > 0, "we only support num_seq_q 1..5 (and a HIP device must be present)")
    return n


def _assign_cpu(num_seq_kvcache, num_head_kv, num_seq_q, new_kv_included, min_process_len, placehold):
    # reference assign_attention_decode_task_cpu_entry (entry.cc:727-778)
    _C.require(num_seq_kvcache.device.type == "cpu", "num_seq_kvcache tensor must be cpu")
    _C.require(num_seq_kvcache.dtype == torch.int32, "num_seq_kvcache dtype must be int32")
    lens = num_seq_kvcache.contiguous()
    bins = num_bins(num_seq_q)
    lp = ctypes.cast(lens.data_ptr(), _INT_P)
    args = (bins, lens.numel(), int(num_head_kv), int(num_seq_q), int(bool(new_kv_included)),
            int(min_process_len))
    rows = _C.lib.hpc_assign_attention_decode_task_rows(lp, *args)
    _C.check(0 if rows > 0 else rows, "assign_attention_decode_task_sync")
    task_map = torch.zeros((rows, 48), dtype=torch.int8)
    rc = _C.lib.hpc_assign_attention_decode_task_sync(
        lp, *args, ctypes.cast(task_map.data_ptr(), _INT_P), rows
    )
    _C.check(0 if rc == rows else (rc if rc < 0 else -2), "assign_attention_decode_task_sync")
    return task_map


def _assign_cuda(num_seq_kvcache, num_head_kv, num_seq_q, new_kv_included, min_process_len, task_map):
    # reference assign_attention_decode_task_cuda_entry (entry.cc:780-817)
    _C.require(num_seq_kvcache.is_cuda, "num_seq_kvcache tensor must be cuda")
    _C.require(num_seq_kvcache.dtype == torch.int32, "num_seq_kvcache dtype must be int32")
    _C.require(num_seq_kvcache.is_contiguous(), "num_seq_kvcache tensor must be contiguous")
    _C.require(task_map is not None, "assign_attention_decode_task_cuda must use task_map output.")
    _C.require(num_seq_kvcache.size(0) <= 4096,
               "assign_attention_decode_task_cuda only support batch_size <= 4096")
    bins = num_bins(num_seq_q, num_seq_kvcache.device)
    rc = _C.lib.hpc_assign_attention_decode_task_async(
        ctypes.cast(task_map.data_ptr(), _INT_P),
        ctypes.cast(num_seq_kvcache.data_ptr(), _INT_P),
        bins, num_seq_kvcache.size(0), int(num_head_kv), int(num_seq_q),
        int(bool(new_kv_included)), int(min_process_len), _C.stream_of(num_seq_kvcache),
    )
    _C.check(rc, "assign_attention_decode_task_async")
    return task_map


_T.impl("assign_attention_decode_task", _assign_cpu, "CPU")
_T.impl("assign_attention_decode_task", _assign_cuda, "CUDA")


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


def _alloc_task_workspace(device, num_cu, max_num_batch, max_seqlen, num_head_kv, min_process_len):
    total, sched = task_workspace_bytes(num_cu, max_num_batch, max_seqlen, num_head_kv, min_process_len)
    ws = torch.zeros(total, dtype=torch.int8, device=device)
    hdr = ws.view(torch.int32)
    hdr[2] = num_head_kv
    hdr[3] = max_num_batch
    hdr[4] = sched
    return ws


_DECODE_WS = {}


def _decode_workspace(device, nbytes):
    """Scratch of a decode call: [arrival counters of split requests | split-KV partials].  The counter region
    (hpc_attention_decode_workspace_zero_bytes(), a fixed place and size) must be zero the first time a buffer is
    used and every call leaves it zero, so the buffer is allocated ZEROED once and then reused: one buffer per
    (device, stream) instead of an allocator round trip on every step (the reference allocates per call and zeroes
    its split_flag per call, src/attention/entry.cc:660-663, 690-694).  Calls on one stream are ordered, so sharing
    the buffer between them is safe; different streams get different buffers.  While a hipGraph is being captured
    the zero-fill of a new buffer is only RECORDED (a node of that graph): such a buffer serves the calls of THAT
    capture and nothing else - the key carries the capture id, the entry of an earlier capture is dropped (its memory
    stays with its graph's pool), and a buffer cached by eager calls is not used inside a capture either, so every
    graph owns its buffer and its zero node and two graphs never share one."""
    stream = torch.cuda.current_stream(device).cuda_stream
    cap = int(_C.lib.hpc_stream_capture_id(c_void_p(stream)))
    _C.require(cap >= 0, "hipStreamGetCaptureInfo failed")
    key = (device.index, stream, cap)
    ws = _DECODE_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if cap:
            for old in [k for k in _DECODE_WS if k[:2] == key[:2] and k[2] not in (0, cap)]:
                del _DECODE_WS[old]
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        ws[: _C.lib.hpc_attention_decode_workspace_zero_bytes()].zero_()
        _DECODE_WS[key] = ws
    return ws


def release_decode_workspaces():
    """Drop the cached decode scratch buffers (e.g. after the streams / graphs that used them are gone) - the Python
    entries' and, when the C++ shim serves the decode ops, its cache too."""
    _DECODE_WS.clear()
    if "attention_decode_fp8" in _C.NATIVE_OPS:
        torch.ops.hpc._release_decode_workspaces()


def _decode_common_checks(q, kcache, vcache, block_ids, num_seq_kvcache, mtp, max_mtp):
    _C.require(q.is_cuda, "q tensor must be cuda")
    _C.require(kcache.is_cuda, "kcache tensor must be cuda")
    _C.require(vcache.is_cuda, "vcache tensor must be cuda")
    _C.require(block_ids.is_cuda, "block_ids tensor must be cuda")
    _C.require(block_ids.is_contiguous(), "block_ids tensor must be contiguous")
    _C.require(num_seq_kvcache.is_contiguous(), "num_seq_kvcache tensor must be contiguous")
    _C.require(block_ids.dtype == torch.int32, "block_ids dtype must be int32")
    _C.require(num_seq_kvcache.dtype == torch.int32, "num_seq_kvcache dtype must be int32")
    _C.require(0 <= mtp <= max_mtp, "we only support mtp 0.." + str(max_mtp) + ".")
    num_batch = num_seq_kvcache.size(0)
    num_seq_q = q.size(0) // num_batch
    _C.require(num_seq_q == mtp + 1, "every request num_seq_q must be mtp + 1")
    _C.require(q.size(2) == 128, "we only support head dim 128.")
    _C.require(q.stride(2) == 1 and q.stride(1) == 128, "q heads must be contiguous")
    _C.require(kcache.stride(3) == 1 and vcache.stride(3) == 1, "kv cache dims must be contiguous")
    heads_per_group = q.size(1) // kcache.size(2)
    _C.require(heads_per_group in (4, 8), "we only support num_head_q / num_head_k == 4 or 8.")
    return num_batch, num_seq_q, heads_per_group


def _schedule_on_the_fly(num_seq_kvcache, block_ids, block_size, num_head_kv, num_seq_q,
                         new_kv_included):
    """No task_map given (reference static split-K path, entry.cc:507-563): build the dynamic
    schedule on the fly with the device scheduler; the page table bounds the sequence length."""
    dev = num_seq_kvcache.device
    num_cu = _C.cu_count(dev)
    max_seq = block_ids.size(1) * block_size + num_seq_q
    tm = _alloc_task_workspace(dev, num_cu, num_seq_kvcache.size(0), max_seq, num_head_kv, 512)
    return _assign_cuda(num_seq_kvcache, num_head_kv, num_seq_q, new_kv_included, 512, tm)


def _attention_decode_bf16_entry(q, kcache, vcache, block_ids, num_seq_kvcache, mtp,
                                 new_kv_included, use_splitk, task_map, split_flag, output):
    num_batch, num_seq_q, group = _decode_common_checks(
        q, kcache, vcache, block_ids, num_seq_kvcache, mtp, 4)
    _C.require(q.dtype == torch.bfloat16, "q dtype must be bfloat16")
    _C.require(kcache.dtype == torch.bfloat16 and vcache.dtype == torch.bfloat16,
               "kv cache dtype must be bfloat16")
    block_size = kcache.size(1)
    _C.require(block_size in (16, 32, 64), "kvcache paged blocksize must be 16, 32 or 64.")
    num_head_q, num_head_kv = q.size(1), kcache.size(2)
    if task_map is not None:
        _C.require(task_map.is_cuda, "task_map tensor must be cuda")
        _C.require(task_map.is_contiguous(), "task_map tensor must be contiguous")
        _C.require(task_map.dtype in (torch.int8, torch.int32),
                   "task_map dtype must be int8 (raw workspace) or int32 (typed)")
        _C.require(use_splitk, "attention_decode_bf16: splitk must be true with a task_map.")
    else:
        task_map = _schedule_on_the_fly(num_seq_kvcache, block_ids, block_size, num_head_kv,
                                        num_seq_q, new_kv_included)
    y = output if output is not None else torch.empty(
        (num_batch * num_seq_q, num_head_q, vcache.size(3)), dtype=torch.bfloat16, device=q.device)
    bins = num_bins(num_seq_q, q.device)
    ws_bytes = _C.lib.hpc_attention_decode_workspace_bytes(bins, num_batch, num_head_kv, num_seq_q, group)
    ws = _decode_workspace(q.device, ws_bytes)
    rc = _C.lib.hpc_attention_decode_bf16_async(
        _C.ptr(y), _C.ptr(ws), ctypes.cast(task_map.data_ptr(), _INT_P), _C.ptr(q), _C.ptr(kcache),
        _C.ptr(vcache), ctypes.cast(block_ids.data_ptr(), _INT_P),
        ctypes.cast(num_seq_kvcache.data_ptr(), _INT_P) if num_seq_kvcache.is_cuda else None, int(bool(new_kv_included)),
        bins, num_batch, num_seq_q,
        num_head_q, num_head_kv, q.size(2), vcache.size(3), block_size, block_ids.size(1),
        y.stride(0), q.stride(0), kcache.stride(0), kcache.stride(1), kcache.stride(2),
        vcache.stride(0), vcache.stride(1), vcache.stride(2), _C.stream_of(q),
    )
    _C.check(rc, "attn decode kernel")
    return y


_T.impl("attention_decode_bf16", _attention_decode_bf16_entry, "CUDA")


def _attention_decode_fp8_entry(q, kcache, vcache, block_ids, num_seq_kvcache, qscale, kscale,
                                vscale, mtp, new_kv_included, quant_type, use_splitk, task_map,
                                split_flag, output):
    # reference attention_decode_fp8_entry, src/attention/entry.cc:569-725
    num_batch, num_seq_q, group = _decode_common_checks(
        q, kcache, vcache, block_ids, num_seq_kvcache, mtp, 3)
    _C.require(q.dtype == torch.float8_e4m3fn, "q dtype must be fp8_e4m3fn")
    _C.require(kcache.element_size() == 1, "kcache tensor element type size must be fp8_e4m3")
    _C.require(vcache.element_size() == 1, "vcache tensor element type size must be fp8_e4m3")
    _C.require(qscale.dtype == torch.float32 and vscale.dtype == torch.float32,
               "qscale / vscale must be float32")
    _C.require(quant_type in (0, 1), "quant_type must be QPERTOKEN_PERHEAD_KPERTOKEN_PERHEAD_VPERHEAD "
               "or QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR")
    _C.require(num_seq_kvcache.is_cuda, "num_seq_kvcache tensor must be cuda")
    block_size = kcache.size(1)
    if quant_type == 0:
        _C.require(block_size in (32, 64), "kvcache paged blocksize must be 32 or 64.")
        _C.require(kscale.dim() == 4 and kscale.stride(3) == 1 and kscale.element_size() == 1,
                   "per-token kscale must be the byte view of the K-cache tail rows")
        ks = (kscale.stride(0), kscale.stride(1), kscale.stride(2))
    else:
        _C.require(block_size in (16, 32, 64), "kvcache paged blocksize must be 16, 32 or 64.")
        _C.require(kscale.dtype == torch.float32 and kscale.numel() >= 1, "kscale must be float32 [1]")
        ks = (0, 0, 0)
    num_head_q, num_head_kv = q.size(1), kcache.size(2)
    if task_map is None:
        task_map = _schedule_on_the_fly(num_seq_kvcache, block_ids, block_size, num_head_kv,
                                        num_seq_q, new_kv_included)
    else:
        _C.require(task_map.is_cuda and task_map.is_contiguous(), "task_map tensor must be cuda, contiguous")
    y = output if output is not None else torch.empty(
        (num_batch * num_seq_q, num_head_q, vcache.size(3)), dtype=torch.bfloat16, device=q.device)
    bins = num_bins(num_seq_q, q.device)
    ws_bytes = _C.lib.hpc_attention_decode_workspace_bytes(bins, num_batch, num_head_kv, num_seq_q, group)
    ws = _decode_workspace(q.device, ws_bytes)
    rc = _C.lib.hpc_attention_decode_fp8_async(
        _C.ptr(y), _C.ptr(ws), ctypes.cast(task_map.data_ptr(), _INT_P), _C.ptr(q), _C.ptr(kcache),
        _C.ptr(vcache), ctypes.cast(block_ids.data_ptr(), _INT_P),
        ctypes.cast(num_seq_kvcache.data_ptr(), _INT_P), _C.ptr(qscale), _C.ptr(kscale),
        _C.ptr(vscale), int(bool(new_kv_included)), int(quant_type), bins, num_batch, num_seq_q, num_head_q, num_head_kv,
        q.size(2), vcache.size(3), block_size, block_ids.size(1), qscale.stride(0), y.stride(0),
        q.stride(0), kcache.stride(0), kcache.stride(1), kcache.stride(2), vcache.stride(0),
        vcache.stride(1), vcache.stride(2), ks[0], ks[1], ks[2], _C.stream_of(q),
    )
    _C.check(rc, "attn decode kernel")
    return y


_T.impl("attention_decode_fp8", _attention_decode_fp8_entry, "CUDA")


# ---------------------------------------------------------------------------- fp8 paged prefill
_T.define(  # verbatim: reference src/attention/entry.cc:835
    "attention_with_kvcache_prefill_fp8(Tensor q, Tensor kcache, Tensor vcache,Tensor qscale, Tensor "
    "kscale, Tensor vscale, Tensor cu_seqlens_q,Tensor block_ids, Tensor num_seq_kvcache, int "
    "max_seqlens_q, int quant_type,Tensor? output) -> (Tensor)"
)


def _attention_prefill_fp8_entry(q, kcache, vcache, qscale, kscale, vscale, cu_seqlens_q, block_ids,
                                 seqlens_kvcache, max_seqlens_q, quant_type, output=None, block_mask=None,
                                 blocksparse=False):
    # reference attention_with_kvcache_prefill_fp8_entry, src/attention/entry.cc:152-262
    for t, name in ((q, "q"), (kcache, "kcache"), (vcache, "vcache"), (qscale, "qscale"), (kscale, "kscale"),
                    (vscale, "vscale"), (cu_seqlens_q, "cu_seqlens_q"), (block_ids, "block_ids"),
                    (seqlens_kvcache, "seqlens_kvcache")):
        _C.require(t.is_cuda, f"{name} tensor must be cuda")
    _C.require(quant_type in (0, 1), "quant_type only support 0/1")
    _C.require(kscale.element_size() in (1, 4), "kscale dtype must be float or fp8")
    _C.require(q.dtype == torch.float8_e4m3fn, "q dtype must be float8_e4m3fn")
    _C.require(kcache.dtype == torch.float8_e4m3fn, "kcache dtype must be float8_e4m3fn")
    _C.require(vcache.dtype == torch.float8_e4m3fn, "vcache dtype must be float8_e4m3fn")
    _C.require(q.dim() == 3 and q.stride(2) == 1 and q.stride(1) == q.size(2), "q must be [total_seq, Hq, D] row-major")
    total_q, num_head_q, dim_qk = q.shape
    num_batch = cu_seqlens_q.size(0) - 1
    block_size, num_head_kv, dim_v = kcache.size(1), kcache.size(2), vcache.size(3)
    _C.require(dim_qk == 128 and dim_v == 128,
               f"attention_with_kvcache_prefill_fp8: expected dim_qk=128 and dim_v=128, got dim_qk={dim_qk} dim_v={dim_v}")
    _C.require(kcache.stride(3) == 1 and vcache.stride(3) == 1, "kv cache head dim must be contiguous")
    _C.require(qscale.dtype == torch.float32 and qscale.dim() == 3 and qscale.is_contiguous()
               and qscale.size(0) == num_batch and qscale.size(1) == num_head_q and qscale.size(2) >= max_seqlens_q,
               "qscale must be float32 [num_batch, num_head_q, max_seqlens_q_pad]")
    for t, name in ((cu_seqlens_q, "cu_seqlens_q"), (block_ids, "block_ids"), (seqlens_kvcache, "seqlens_kvcache")):
        _C.require(t.dtype == torch.int32 and t.is_contiguous(), f"{name} must be contiguous int32")
    _C.require(vscale.dtype == torch.float32, "vscale must be float32")
    if quant_type == 0:
        _C.require(kscale.dim() == 4 and kscale.stride(3) == 1, "per-token kscale must be the K-cache tail rows view")
        es = kscale.element_size()
        ks = (kscale.stride(0) * es, kscale.stride(1) * es, kscale.stride(2) * es)
        _C.require(vscale.numel() >= num_head_kv, "vscale must hold one value per kv head")
    else:
        _C.require(kscale.dtype == torch.float32 and kscale.numel() >= 1, "kscale must be float32 [1]")
        ks = (0, 0, 0)
    if output is not None:
        _C.require(output.is_cuda and output.device == q.device, "output tensor must be on the same device as q")
        _C.require(output.dtype == torch.bfloat16, "output dtype must be bfloat16")
        _C.require(output.is_contiguous(), "output tensor must be contiguous")
        _C.require(tuple(output.shape) == (total_q, num_head_q, dim_v),
                   "output must have shape [total_seq_q, num_head_q, num_dim_v]")
        y = output
    else:
        y = torch.empty((total_q, num_head_q, dim_v), dtype=torch.bfloat16, device=q.device)
    if total_q == 0:
        return y
    if blocksparse:
        _C.require(128 % block_size == 0, "unsupported block_size for FP8 blocksparse prefill")
        tiles_m = (int(max_seqlens_q) + 127) // 128
        tiles_kv = 0
        if block_mask is not None:
            _C.require(block_mask.device == q.device, "block_mask tensor must be on the same device as q")
            _C.require(block_mask.dtype == torch.uint8, "block_mask dtype must be uint8")
            _C.require(block_mask.is_contiguous(), "block_mask tensor must be contiguous")
            _C.require(block_mask.dim() == 4 and block_mask.size(0) == num_batch and block_mask.size(1) == num_head_q
                       and block_mask.size(2) == tiles_m,
                       f"block_mask must have shape [{num_batch}, {num_head_q}, {tiles_m}, Kb] where "
                       "Kb = ceil(max_kv_len / kTileN=128)")
            tiles_kv = block_mask.size(3)
            _C.require(tiles_kv > 0, "block_mask Kb dim must be > 0")
        rc = _C.lib.hpc_attention_with_kvcache_blocksparse_prefill_fp8_async(
            _C.ptr(y), _C.ptr(q), _C.ptr(kcache), _C.ptr(vcache), _C.ptr(qscale), _C.ptr(kscale), _C.ptr(vscale),
            _C.ptr(cu_seqlens_q), _C.ptr(block_ids), _C.ptr(seqlens_kvcache), _C.ptr(block_mask), tiles_m, tiles_kv,
            int(quant_type), num_batch, int(max_seqlens_q), qscale.size(2), dim_qk, dim_v, num_head_q, num_head_kv,
            block_size, block_ids.size(1), y.stride(0), q.stride(0), kcache.stride(0), kcache.stride(1),
            kcache.stride(2), vcache.stride(0), vcache.stride(1), vcache.stride(2), ks[0], ks[1], ks[2],
            _C.stream_of(q))
        _C.check(rc, "attention_with_kvcache_blocksparse_prefill_fp8")
        return y
    rc = _C.lib.hpc_attention_with_kvcache_prefill_fp8_async(
        _C.ptr(y), _C.ptr(q), _C.ptr(kcache), _C.ptr(vcache), _C.ptr(qscale), _C.ptr(kscale), _C.ptr(vscale),
        _C.ptr(cu_seqlens_q), _C.ptr(block_ids), _C.ptr(seqlens_kvcache), int(quant_type), num_batch,
        int(max_seqlens_q), qscale.size(2), dim_qk, dim_v, num_head_q, num_head_kv, block_size, block_ids.size(1),
        y.stride(0), q.stride(0), kcache.stride(0), kcache.stride(1), kcache.stride(2), vcache.stride(0),
        vcache.stride(1), vcache.stride(2), ks[0], ks[1], ks[2], _C.stream_of(q))
    _C.check(rc, "attention_with_kvcache_prefill_fp8")
    return y


_T.impl("attention_with_kvcache_prefill_fp8", _attention_prefill_fp8_entry, "CUDA")


_T.define(  # verbatim: reference src/attention/entry.cc:843
    "attention_with_kvcache_blocksparse_prefill_fp8(Tensor q, Tensor kcache, Tensor vcache,Tensor qscale,"
    " Tensor kscale, Tensor vscale, Tensor cu_seqlens_q,Tensor block_ids, Tensor num_seq_kvcache, int "
    "max_seqlens_q, int quant_type,Tensor? block_mask, Tensor? output) -> (Tensor)"
)


def _attention_blocksparse_prefill_fp8_entry(q, kcache, vcache, qscale, kscale, vscale, cu_seqlens_q, block_ids,
                                             seqlens_kvcache, max_seqlens_q, quant_type, block_mask=None,
                                             output=None):
    # reference attention_with_kvcache_blocksparse_prefill_fp8_entry, src/attention/entry.cc:264-409
    return _attention_prefill_fp8_entry(q, kcache, vcache, qscale, kscale, vscale, cu_seqlens_q, block_ids,
                                        seqlens_kvcache, max_seqlens_q, quant_type, output, block_mask, True)


_T.impl("attention_with_kvcache_blocksparse_prefill_fp8", _attention_blocksparse_prefill_fp8_entry, "CUDA")


# ---------------------------------------------------------------------------- bf16 prefill
_T.define("attention_prefill_bf16(Tensor q, Tensor k, Tensor v, Tensor seqlens_q, Tensor cu_seqlens_q, "
          "int max_seqlens_q, Tensor? output) -> (Tensor)")
_T.define("attention_with_kvcache_prefill_bf16(Tensor q, Tensor kcache, Tensor vcache,"
          "Tensor cu_seqlens_q, "
          "Tensor block_ids, Tensor num_seq_kvcache, int max_seqlens_q, Tensor? output) -> (Tensor)")


def _prefill_output(q, output, dim_v):
    total_q, num_head_q = q.size(0), q.size(1)
    if output is None:
        return torch.empty((total_q, num_head_q, dim_v), dtype=torch.bfloat16, device=q.device)
    _C.require(output.is_cuda and output.device == q.device, "output tensor must be on the same device as q")
    _C.require(output.dtype == torch.bfloat16, "output dtype must be bfloat16")
    _C.require(output.is_contiguous(), "output tensor must be contiguous")
    _C.require(tuple(output.shape) == (total_q, num_head_q, dim_v),
               "output must have shape [total_seq_q, num_head_q, num_dim_v]")
    return output


def _attention_prefill_bf16_entry(q, k, v, seqlens_q, cu_seqlens_q, max_seqlens_q, output=None):
    # reference attention_prefill_bf16_entry, src/attention/entry.cc:15-81
    for t, name in ((q, "q"), (k, "k"), (v, "v"), (seqlens_q, "seqlens_q"), (cu_seqlens_q, "cu_seqlens_q")):
        _C.require(t.is_cuda, f"{name} tensor must be cuda")
    for t, name in ((q, "q"), (k, "k"), (v, "v")):
        _C.require(t.dtype == torch.bfloat16 and t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == t.size(2),
                   f"{name} must be bfloat16 [total_seq, heads, dim] with contiguous heads")
    _C.require(cu_seqlens_q.dtype == torch.int32 and cu_seqlens_q.is_contiguous(), "cu_seqlens_q must be contiguous int32")
    _C.require(q.size(2) == 128 and k.size(2) == 128 and v.size(2) == 128,
               "attention_prefill_bf16: expected dim_qk=128 and dim_v=128")
    _C.require(k.size(0) == q.size(0) and v.size(0) == q.size(0) and k.size(1) == v.size(1),
               "k / v must hold one row per q token and the same number of kv heads")
    y = _prefill_output(q, output, v.size(2))
    if q.size(0) == 0:
        return y
    rc = _C.lib.hpc_attention_prefill_bf16_async(
        _C.ptr(y), _C.ptr(q), _C.ptr(k), _C.ptr(v), _C.ptr(cu_seqlens_q), cu_seqlens_q.size(0) - 1, int(max_seqlens_q),
        128, 128, q.size(1), k.size(1), y.stride(0), q.stride(0), k.stride(0), v.stride(0), _C.stream_of(q))
    _C.check(rc, "attention_prefill_bf16")
    return y


def _attention_kvcache_prefill_bf16_entry(q, kcache, vcache, cu_seqlens_q, block_ids, num_seq_kvcache, max_seqlens_q,
                                          output=None):
    # reference attention_with_kvcache_prefill_bf16_entry, src/attention/entry.cc:83-150
    for t, name in ((q, "q"), (kcache, "kcache"), (vcache, "vcache"), (cu_seqlens_q, "cu_seqlens_q"),
                    (block_ids, "block_ids"), (num_seq_kvcache, "seqlens_kvcache")):
        _C.require(t.is_cuda, f"{name} tensor must be cuda")
    for t, name in ((q, "q"), (kcache, "kcache"), (vcache, "vcache")):
        _C.require(t.dtype == torch.bfloat16, f"{name} dtype must be bfloat16")
    _C.require(q.dim() == 3 and q.stride(2) == 1 and q.stride(1) == q.size(2), "q must be [total_seq, Hq, D] row-major")
    dim_qk, dim_v = q.size(2), vcache.size(3)
    _C.require(dim_qk == 128 and dim_v == 128,
               f"attention_with_kvcache_prefill_bf16: expected dim_qk=128 and dim_v=128, got dim_qk={dim_qk} dim_v={dim_v}")
    _C.require(kcache.stride(3) == 1 and vcache.stride(3) == 1, "kv cache head dim must be contiguous")
    for t, name in ((cu_seqlens_q, "cu_seqlens_q"), (block_ids, "block_ids"), (num_seq_kvcache, "seqlens_kvcache")):
        _C.require(t.dtype == torch.int32 and t.is_contiguous(), f"{name} must be contiguous int32")
    y = _prefill_output(q, output, dim_v)
    if q.size(0) == 0:
        return y
    rc = _C.lib.hpc_attention_with_kvcache_prefill_bf16_async(
        _C.ptr(y), _C.ptr(q), _C.ptr(kcache), _C.ptr(vcache), _C.ptr(cu_seqlens_q), _C.ptr(block_ids),
        _C.ptr(num_seq_kvcache), cu_seqlens_q.size(0) - 1, int(max_seqlens_q), dim_qk, dim_v, q.size(1), kcache.size(2),
        kcache.size(1), block_ids.size(1), y.stride(0), q.stride(0), kcache.stride(0), kcache.stride(1),
        kcache.stride(2), vcache.stride(0), vcache.stride(1), vcache.stride(2), _C.stream_of(q))
    _C.check(rc, "attention_with_kvcache_prefill_bf16")
    return y


_T.impl("attention_prefill_bf16", _attention_prefill_bf16_entry, "CUDA")
_T.impl("attention_with_kvcache_prefill_bf16", _attention_kvcache_prefill_bf16_entry, "CUDA")
