"""torch.ops.hpc.rope_norm_store_kv[_fp8] (reference src/rope/entry.cc:14-240: same schemas, checks and
output allocation); compute in csrc/rope.hip."""
import torch

from . import _C

_T = _C.torch_lib
_F8 = torch.float8_e4m3fn
_T.define(
    "rope_norm_store_kv(Tensor! kcache, Tensor! vcache, Tensor qkv, Tensor cos_sin, "
    "Tensor num_seqlen_per_req, Tensor q_index, Tensor kvcache_indices, bool is_prefill, "
    "Tensor? q_norm_weight, Tensor? k_norm_weight, "
    "Tensor? out_q=None, Tensor? out_k=None, Tensor? out_v=None, int qk_norm_policy=0) -> Tensor"
)
_T.define(  # verbatim: reference src/rope/entry.cc:231
    "rope_norm_store_kv_fp8(Tensor! kcache, Tensor! vcache, Tensor qkv, Tensor cos_sin, Tensor "
    "num_seqlen_per_req, Tensor q_index, Tensor kvcache_indices, bool is_prefill, Tensor k_scale, Tensor "
    "v_scale, int quant_policy, int max_seqlens, float? upper_max, Tensor? q_scale_inv, Tensor? "
    "q_norm_weight, Tensor? k_norm_weight, Tensor? out_q=None, Tensor? out_k=None, Tensor? out_v=None, "
    "int qk_norm_policy=0) -> (Tensor, Tensor, Tensor)"
)


def _common(kcache, vcache, qkv, cos_sin, num_seqlen_per_req, q_index, kvcache_indices, q_norm_weight,
            k_norm_weight, qk_norm_policy):
    _C.require(qkv.is_cuda and qkv.is_contiguous(), "qkv tensor must be contiguous")
    _C.require(cos_sin.is_contiguous() and cos_sin.dtype == torch.float32, "cos_sin tensor must be contiguous float32")
    _C.require(num_seqlen_per_req.is_contiguous() and num_seqlen_per_req.dtype == torch.int32,
               "num_seqlen_per_req tensor must be contiguous int32")
    _C.require(q_index.is_contiguous() and q_index.dtype == torch.int32, "q_index must be contiguous int32")
    _C.require(kvcache_indices.is_contiguous() and kvcache_indices.dtype == torch.int32,
               "kvcache_indices tensor must be contiguous int32")
    _C.require(0 <= qk_norm_policy <= 2, "qk_norm_policy must be 0, 1 or 2")
    _C.require(qkv.dtype == torch.bfloat16, "qkv must be bfloat16")
    num_kv, qk_dim, v_dim = kcache.size(2), kcache.size(3), vcache.size(3)
    _C.require(qk_dim == 128 and v_dim == 128, "head dims must be 128")
    hidden = qkv.size(1)
    num_q = (hidden - num_kv * qk_dim - num_kv * v_dim) // qk_dim
    _C.require(num_q > 0 and (num_q + 2 * num_kv) * qk_dim == hidden, "qkv hidden size does not match the caches")
    for c in (kcache, vcache):
        _C.require(c.stride(3) == 1 and c.stride(2) == 128 and c.stride(1) == num_kv * 128,
                   "kv cache pages must be [block_size, num_kv_heads, 128] contiguous")
    for wt in (q_norm_weight, k_norm_weight):
        if wt is not None:
            _C.require(wt.dtype == torch.float32 and wt.numel() == 128, "norm weights must be float32 [128]")
    if qk_norm_policy:
        _C.require(q_norm_weight is not None and k_norm_weight is not None,
                   "q_norm_weight / k_norm_weight are required when qk_norm_policy != 0")
    return num_q, num_kv


def _rope_entry(kcache, vcache, qkv, cos_sin, num_seqlen_per_req, q_index, kvcache_indices, is_prefill,
                q_norm_weight=None, k_norm_weight=None, out_q=None, out_k=None, out_v=None, qk_norm_policy=0):
    num_q, num_kv = _common(kcache, vcache, qkv, cos_sin, num_seqlen_per_req, q_index, kvcache_indices,
                            q_norm_weight, k_norm_weight, qk_norm_policy)
    _C.require(kcache.dtype == torch.bfloat16 and vcache.dtype == torch.bfloat16, "caches must be bfloat16")
    rows = qkv.size(0)
    if out_q is None:
        out_q = torch.empty((rows, num_q, 128), dtype=torch.bfloat16, device=qkv.device)
    else:
        _C.require(out_q.is_contiguous(), "out_q tensor must be contiguous")
    for t, n in ((out_k, "out_k"), (out_v, "out_v")):
        if t is not None:
            _C.require(t.is_contiguous(), f"{n} tensor must be contiguous")
    rc = _C.lib.hpc_rope_norm_store_kv_async(
        _C.ptr(out_q), _C.ptr(kcache), _C.ptr(vcache), _C.ptr(out_k), _C.ptr(out_v), _C.ptr(qkv), _C.ptr(cos_sin),
        _C.ptr(num_seqlen_per_req), _C.ptr(q_index), _C.ptr(kvcache_indices), _C.ptr(q_norm_weight),
        _C.ptr(k_norm_weight), kcache.stride(0), vcache.stride(0), num_seqlen_per_req.size(0),
        kvcache_indices.size(1), kcache.size(1), rows, num_q, num_kv, 128, 128, int(bool(is_prefill)),
        int(qk_norm_policy), _C.stream_of(qkv))
    _C.check(rc, "rope_norm_store_kv_async")
    return out_q


_T.impl("rope_norm_store_kv", _rope_entry, "CUDA")


def _rope_fp8_entry(kcache, vcache, qkv, cos_sin, num_seqlen_per_req, q_index, kvcache_indices, is_prefill,
                    k_scale, v_scale, quant_policy, max_seqlens, upper_max=None, q_scale_inv=None,
                    q_norm_weight=None, k_norm_weight=None, out_q=None, out_k=None, out_v=None, qk_norm_policy=0):
    num_q, num_kv = _common(kcache, vcache, qkv, cos_sin, num_seqlen_per_req, q_index, kvcache_indices,
                            q_norm_weight, k_norm_weight, qk_norm_policy)
    _C.require(k_scale.dim() == 1 and k_scale.size(0) == 1, "k_scale must contain 1 element")
    _C.require(v_scale.dim() == 1 and v_scale.size(0) == 1, "v_scale must contain 1 element")
    _C.require(quant_policy in (1, 2), "quant_policy must be 1 or 2")
    _C.require(kcache.element_size() == 1 and vcache.element_size() == 1, "caches must be 1-byte dtype")
    fp8_max = 448.0
    if upper_max is not None:
        _C.require(not upper_max > fp8_max, "upper_max should not be larger than fp8_max")
        fp8_max = float(upper_max)
    rows, num_req, dev = qkv.size(0), num_seqlen_per_req.size(0), qkv.device
    if out_q is None:
        out_q = torch.empty((rows, num_q, 128), dtype=_F8, device=dev)
    else:
        _C.require(out_q.is_contiguous() and out_q.dtype == _F8, "out_q must be contiguous float8_e4m3fn")
    q_scale, pad = None, 0
    if quant_policy == 1:
        if is_prefill:
            pad = (int(max_seqlens) + 127) // 128 * 128
            q_scale = torch.empty((num_req, num_q, pad), dtype=torch.float32, device=dev)
        else:
            q_scale = torch.empty((rows, num_q), dtype=torch.float32, device=dev)
    else:
        _C.require(q_scale_inv is not None and q_scale_inv.dtype == torch.float32,
                   "q_scale_inv required for quant_policy=2")
    split_k_flag = torch.empty((num_req, num_kv), dtype=torch.int32, device=dev)
    for t, n in ((out_k, "out_k"), (out_v, "out_v")):
        if t is not None:
            _C.require(t.is_contiguous() and t.dtype == _F8, f"{n} must be contiguous float8_e4m3fn")
    rc = _C.lib.hpc_rope_norm_store_kv_fp8_async(
        _C.ptr(out_q), _C.ptr(kcache), _C.ptr(vcache), _C.ptr(out_k), _C.ptr(out_v), _C.ptr(split_k_flag),
        _C.ptr(q_scale), _C.ptr(qkv), _C.ptr(cos_sin), _C.ptr(num_seqlen_per_req), _C.ptr(q_index),
        _C.ptr(kvcache_indices), _C.ptr(q_norm_weight), _C.ptr(k_norm_weight), _C.ptr(k_scale), _C.ptr(v_scale),
        _C.ptr(q_scale_inv), fp8_max, pad, kcache.stride(0), vcache.stride(0), num_req, kvcache_indices.size(1),
        kcache.size(1), rows, num_q, num_kv, 128, 128, int(bool(is_prefill)), int(qk_norm_policy),
        int(quant_policy), _C.stream_of(qkv))
    _C.check(rc, "rope_norm_store_kv_fp8_async")
    return out_q, q_scale, split_k_flag


_T.impl("rope_norm_store_kv_fp8", _rope_fp8_entry, "CUDA")
