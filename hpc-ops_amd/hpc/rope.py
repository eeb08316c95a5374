"""hpc.rope — RoPE + QK-norm + paged KV store (reference hpc/rope.py:7-234)."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _C  # noqa: F401  (loads the libraries that register torch.ops.hpc.*)


def rope_norm_store_kv(
    key_cache: Tensor,
    value_cache: Tensor,
    qkv: Tensor,
    cos_sin: Tensor,
    num_seqlen_per_req: Tensor,
    q_index: Tensor,
    kvcache_indices: Tensor,
    is_prefill: bool,
    q_norm_weight: Optional[Tensor] = None,
    k_norm_weight: Optional[Tensor] = None,
    out_q: Optional[Tensor] = None,
    out_k: Optional[Tensor] = None,
    out_v: Optional[Tensor] = None,
    qk_norm_policy: int = 0,
) -> Tensor:
    """RoPE on Q/K (neox pairing), optional QK RMSNorm (policy 1: after RoPE, 2: before), K/V written
    into the paged bf16 KV cache (or out_k / out_v), tail of each request's last page zeroed.
    Shapes as in the reference (hpc/rope.py:7-104); head dims 128.  Returns Q [rows, Hq, 128] bf16."""
    return torch.ops.hpc.rope_norm_store_kv(
        key_cache, value_cache, qkv, cos_sin, num_seqlen_per_req, q_index, kvcache_indices, is_prefill,
        q_norm_weight, k_norm_weight, out_q, out_k, out_v, qk_norm_policy,
    )


def rope_norm_store_kv_fp8(
    key_cache: Tensor,
    value_cache: Tensor,
    qkv: Tensor,
    cos_sin: Tensor,
    num_seqlen_per_req: Tensor,
    q_index: Tensor,
    kvcache_indices: Tensor,
    is_prefill: bool,
    k_scale: Tensor,
    v_scale: Tensor,
    quant_policy: int,
    max_seqlens: int = 0,
    upper_max: Optional[float] = None,
    q_scale_inv: Optional[Tensor] = None,
    q_norm_weight: Optional[Tensor] = None,
    k_norm_weight: Optional[Tensor] = None,
    out_q: Optional[Tensor] = None,
    out_k: Optional[Tensor] = None,
    out_v: Optional[Tensor] = None,
    qk_norm_policy: int = 0,
) -> Tuple[Tensor, Tensor, Tensor]:
    """FP8 variant (reference hpc/rope.py:107-234): Q quantised per token per head (quant_policy 1,
    scale = amax/upper_max returned in q_scale: decode [rows, Hq], prefill [num_req, Hq, pad128]) or
    statically (quant_policy 2, q_scale_inv); K/V divided by the static k_scale / v_scale into the
    e4m3 paged cache.  Returns (q e4m3, q_scale or None, split_k_flag int32 [num_req, Hkv] zeroed)."""
    return torch.ops.hpc.rope_norm_store_kv_fp8(
        key_cache, value_cache, qkv, cos_sin, num_seqlen_per_req, q_index, kvcache_indices, is_prefill,
        k_scale, v_scale, quant_policy, max_seqlens, upper_max, q_scale_inv, q_norm_weight, k_norm_weight,
        out_q, out_k, out_v, qk_norm_policy,
    )


def _rope_q_heads(key_cache, value_cache, qkv):
    kv_heads, qk_dim, v_dim = key_cache.size(-2), key_cache.size(-1), value_cache.size(-1)
    return (qkv.size(-1) - kv_heads * qk_dim - kv_heads * v_dim) // qk_dim, kv_heads, qk_dim


@torch.library.register_fake("hpc::rope_norm_store_kv")
def _rope_norm_store_kv_fake(key_cache, value_cache, qkv, cos_sin, num_seqlen_per_req, q_index, kvcache_indices, is_prefill,
                             q_norm_weight, k_norm_weight, out_q=None, out_k=None, out_v=None, qk_norm_policy=0):
    if out_q is not None:
        return out_q
    q_heads, _, qk_dim = _rope_q_heads(key_cache, value_cache, qkv)
    return torch.empty((qkv.size(0), q_heads, qk_dim), dtype=qkv.dtype, device=qkv.device)


@torch.library.register_fake("hpc::rope_norm_store_kv_fp8")
def _rope_norm_store_kv_fp8_fake(key_cache, value_cache, qkv, cos_sin, num_seqlen_per_req, q_index, kvcache_indices, is_prefill,
                                 k_scale, v_scale, quant_policy, max_seqlens, upper_max, q_scale_inv, q_norm_weight,
                                 k_norm_weight, out_q=None, out_k=None, out_v=None, qk_norm_policy=0):
    """(q e4m3, q_scale, split_k_flag) with the entry's shapes (csrc/torch_misc.cpp::rope_norm_store_kv_fp8): dynamic q
    scales are [rows, Hq] in decode and [num_req, Hq, pad128(max_seqlens)] in prefill; with static q scales
    (quant_policy 2) the real op returns an undefined tensor there - a fake must return a tensor: an empty one"""
    q_heads, kv_heads, qk_dim = _rope_q_heads(key_cache, value_cache, qkv)
    rows, num_req, dev = qkv.size(0), num_seqlen_per_req.size(0), qkv.device
    q = out_q if out_q is not None else torch.empty((rows, q_heads, qk_dim), dtype=torch.float8_e4m3fn, device=dev)
    if quant_policy == 1:
        shape = (num_req, q_heads, (max_seqlens + 127) // 128 * 128) if is_prefill else (rows, q_heads)
    else:
        shape = (0,)
    return q, torch.empty(shape, dtype=torch.float32, device=dev), torch.empty((num_req, kv_heads), dtype=torch.int32, device=dev)
