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
