"""Decode-attention oracles — TEST INFRASTRUCTURE (see oracle/__init__.py).

CPU restatements of the PyTorch-eager references embedded in the reference's tests:
  ref_attn_with_paged_kvcache        tests/test_attention_decode_bf16.py:15-59
  ref_attn_fp8_kvpertensor           tests/test_attention_decode_qpertoken_perhead_kvpertensor_fp8.py:14-79
Inputs are CPU tensors; math is fp32 like the reference's.
"""
import math

import torch
import torch.nn.functional as F


def ref_attn_with_paged_kvcache(q, kvcache, block_ids, nblocks, num_seq_q, num_seq_kvcache):
    """q [B*Sq, Hq, D] bf16; kvcache [nblk, 2, P, Hkv, D]; num_seq_kvcache = tokens BEFORE the Sq
    new ones.  Returns [B*Sq, Hq, D] in q's dtype.  (reference tests/test_attention_decode_bf16.py:15-59)"""
    num_batch = num_seq_kvcache.shape[0]
    num_head_q, head_dim = q.shape[1], q.shape[2]
    num_head_kv = kvcache.shape[3]
    group = num_head_q // num_head_kv
    qb = q.reshape(num_batch, -1, num_head_q, head_dim)
    out = torch.empty_like(qb)
    for bi in range(num_batch):
        sq = num_seq_q
        q_batch = qb[bi].transpose(0, 1).float()
        blk = block_ids[bi, : int(nblocks[bi])].long()
        seqlen = sq + int(num_seq_kvcache[bi])
        k_batch = (kvcache[blk, 0].reshape(-1, num_head_kv, head_dim).transpose(0, 1)[:, :seqlen]
                   .repeat_interleave(group, dim=0)).float()
        v_batch = (kvcache[blk, 1].reshape(-1, num_head_kv, head_dim).transpose(0, 1)[:, :seqlen]
                   .repeat_interleave(group, dim=0)).float()
        p = q_batch @ k_batch.transpose(-1, -2) / math.sqrt(head_dim)
        causal = torch.cat(
            [torch.ones(sq, seqlen - sq, dtype=torch.bool),
             torch.tril(torch.ones(sq, sq, dtype=torch.bool))], dim=-1).unsqueeze(0)
        p = p.masked_fill(~causal, float("-inf"))
        y = torch.matmul(F.softmax(p, dim=-1), v_batch)
        out[bi] = y.transpose(0, 1).to(out.dtype)
    return out.reshape(-1, num_head_q, head_dim)


def ref_attn_paged_separate(q, k_cache, v_cache, block_ids, kv_lens_total, num_seq_q, rows=None):
    """Same oracle for separate K / V caches [nblk, P, Hkv, D] and lens that already include the
    Sq new tokens (the reference benchmark generator's layout,
    benchmark/attention_decode/bench_attention_decode_bf16.py:125-154).  `rows` restricts the
    computation to a subset of requests (bounded CPU-baseline sample)."""
    num_batch = kv_lens_total.shape[0]
    num_head_q, head_dim = q.shape[1], q.shape[2]
    P, num_head_kv = k_cache.shape[1], k_cache.shape[2]
    group = num_head_q // num_head_kv
    qb = q.reshape(num_batch, num_seq_q, num_head_q, head_dim)
    rows = range(num_batch) if rows is None else rows
    outs = []
    for bi in rows:
        seqlen = int(kv_lens_total[bi])
        nb = (seqlen + P - 1) // P
        blk = block_ids[bi, :nb].long()
        qf = qb[bi].transpose(0, 1).float()
        kf = (k_cache[blk].reshape(-1, num_head_kv, head_dim).transpose(0, 1)[:, :seqlen]
              .repeat_interleave(group, dim=0)).float()
        vf = (v_cache[blk].reshape(-1, num_head_kv, head_dim).transpose(0, 1)[:, :seqlen]
              .repeat_interleave(group, dim=0)).float()
        p = qf @ kf.transpose(-1, -2) / math.sqrt(head_dim)
        sq = num_seq_q
        causal = torch.cat(
            [torch.ones(sq, seqlen - sq, dtype=torch.bool),
             torch.tril(torch.ones(sq, sq, dtype=torch.bool))], dim=-1).unsqueeze(0)
        p = p.masked_fill(~causal, float("-inf"))
        y = torch.matmul(F.softmax(p, dim=-1), vf)
        outs.append(y.transpose(0, 1).to(q.dtype))
    return torch.stack(outs, 0)


# ---- FP8 decode oracles --------------------------------------------------------------------------
def quant_paged_cache_pertoken(cache, block_size):
    """reference tests/test_attention_decode_qkpertoken_perhead_vperhead_fp8.py:14-36.
    cache [nblk, P + P*4/128, Hkv, 128] (float): returns (fp8 cache incl. scale rows, scale view)."""
    num_blocks, head_dim, num_head_kv = cache.shape[0], cache.shape[-1], cache.shape[-2]
    scale = cache[:, :block_size].float().abs().max(-1)[0] / 448
    cache_fp8 = torch.empty_like(cache, dtype=torch.float8_e4m3fn)
    cache_fp8[:, :block_size] = (cache[:, :block_size] / scale[:, :, :, None]).to(torch.float8_e4m3fn)
    scale = (scale.permute(0, 2, 1).contiguous().view(torch.float8_e4m3fn)
             .reshape(num_blocks, num_head_kv, -1, head_dim).permute(0, 2, 1, 3).contiguous())
    cache_fp8[:, block_size:] = scale
    return cache_fp8, cache_fp8[:, block_size:]


def quant_paged_cache_perhead(cache, block_size):
    """reference tests/test_attention_decode_qkpertoken_perhead_vperhead_fp8.py:38-50."""
    num_head_kv = cache.shape[-2]
    scale = (cache[:, :block_size].float().abs().permute(2, 0, 1, 3).reshape(num_head_kv, -1)
             .max(-1)[0] / 448)
    cache_fp8 = (cache.float() / scale[None, None, :, None]).to(torch.float8_e4m3fn)
    return cache_fp8, scale * 0.1


def ref_attn_fp8(q, kvcache, block_ids, nblocks, num_seq_q, num_seq_kvcache, q_scale, k_scale,
                 v_scale, k_per_token, literal_qscale_row=False):
    """FP8 decode oracle.  k_per_token False: reference
    tests/test_attention_decode_qpertoken_perhead_kvpertensor_fp8.py:14-79 (k_scale, v_scale [1]);
    True: tests/test_attention_decode_qkpertoken_perhead_vperhead_fp8.py:53-133 (k_scale = byte view
    of the cache tail rows [nblk, rows, Hkv, 128], v_scale [Hkv]).
    q e4m3 [B*Sq, Hq, D]; kvcache e4m3 [nblk, 2, P, Hkv, D] (token rows only).
    The reference test indexes q_scale[bi] (row bi of [B*Sq, Hq]) for every q row of request bi -
    right only for Sq == 1 and hidden by its tolerance; the kernels use row bi*Sq+s
    (reference ..._kernels.cuh:280-293).  Default here is the kernel's indexing;
    literal_qscale_row=True reproduces the test's."""
    num_batch = num_seq_kvcache.shape[0]
    num_head_q, head_dim = q.shape[1], q.shape[2]
    num_head_kv = kvcache.shape[3]
    group = num_head_q // num_head_kv
    qb = q.reshape(num_batch, -1, num_head_q, head_dim)
    qs = q_scale.reshape(num_batch, -1, num_head_q)
    out = torch.empty(qb.shape, dtype=torch.bfloat16)
    for bi in range(num_batch):
        sq = num_seq_q
        q_batch = qb[bi].transpose(0, 1).float()
        blk = block_ids[bi, : int(nblocks[bi])].long()
        seqlen = sq + int(num_seq_kvcache[bi])
        k_batch = (kvcache[blk, 0].reshape(-1, num_head_kv, head_dim).transpose(0, 1)[:, :seqlen]
                   .repeat_interleave(group, dim=0)).float()
        v_batch = (kvcache[blk, 1].reshape(-1, num_head_kv, head_dim).transpose(0, 1)[:, :seqlen]
                   .repeat_interleave(group, dim=0)).float()
        p = q_batch @ k_batch.transpose(-1, -2) / math.sqrt(head_dim)
        if literal_qscale_row:
            p = p * q_scale[bi][:, None, None]
        else:
            p = p * qs[bi].transpose(0, 1)[:, :, None]  # [Hq, Sq, 1]
        if k_per_token:
            ksb = (k_scale[blk].contiguous().view(torch.float32).permute(0, 1, 3, 2)
                   .reshape(-1, num_head_kv).transpose(0, 1)[:, :seqlen]
                   .repeat_interleave(group, dim=0)).float()
            p = p * ksb.unsqueeze(1)
        else:
            p = p * k_scale
        causal = torch.cat(
            [torch.ones(sq, seqlen - sq, dtype=torch.bool),
             torch.tril(torch.ones(sq, sq, dtype=torch.bool))], dim=-1).unsqueeze(0)
        p = p.masked_fill(~causal, float("-inf"))
        w = torch.exp(p - p.max(dim=-1)[0][:, :, None])
        gsum = w.sum(dim=-1)[:, :, None]
        w = (w * 256.0).to(torch.float8_e4m3fn).float()
        y = torch.matmul(w, v_batch) / gsum
        if k_per_token:
            y = y * v_scale[:, None, None].repeat_interleave(group, dim=0) / 256.0
        else:
            y = y * (v_scale / 256.0)
        out[bi] = y.transpose(0, 1).to(torch.bfloat16)
    return out.reshape(-1, num_head_q, head_dim)


def ref_attn_fp8_separate(q, k_cache, v_cache, block_ids, kv_lens_total, num_seq_q, q_scale, k_scale,
                          v_scale, rows=None):
    """ref_attn_fp8 (per-tensor K/V scales) for separate K / V caches [nblk, P, Hkv, D] and lens that
    already include the Sq new tokens - the layout of the reference benchmark generator
    (benchmark/attention_decode/bench_attention_decode_fp8.py:137-180).  `rows` restricts the
    computation to a subset of requests (graded-shape parity tests and the bounded CPU-baseline
    sample of bench.py).  Same arithmetic as ref_attn_fp8: global-max softmax, P = e4m3(256 p),
    y = P V / sum(p) * vscale / 256.  Returns bf16 [len(rows), Sq, Hq, D]."""
    num_batch = kv_lens_total.shape[0]
    num_head_q, head_dim = q.shape[1], q.shape[2]
    P, num_head_kv = k_cache.shape[1], k_cache.shape[2]
    group = num_head_q // num_head_kv
    qb = q.reshape(num_batch, num_seq_q, num_head_q, head_dim)
    qs = q_scale.reshape(num_batch, num_seq_q, num_head_q)
    rows = range(num_batch) if rows is None else rows
    outs = []
    for bi in rows:
        sq = num_seq_q
        seqlen = int(kv_lens_total[bi])
        blk = block_ids[bi, : (seqlen + P - 1) // P].long()
        q_batch = qb[bi].transpose(0, 1).float()
        k_batch = (k_cache[blk].reshape(-1, num_head_kv, head_dim).transpose(0, 1)[:, :seqlen]
                   .repeat_interleave(group, dim=0)).float()
        v_batch = (v_cache[blk].reshape(-1, num_head_kv, head_dim).transpose(0, 1)[:, :seqlen]
                   .repeat_interleave(group, dim=0)).float()
        p = q_batch @ k_batch.transpose(-1, -2) / math.sqrt(head_dim)
        p = p * qs[bi].transpose(0, 1)[:, :, None] * k_scale
        causal = torch.cat(
            [torch.ones(sq, seqlen - sq, dtype=torch.bool),
             torch.tril(torch.ones(sq, sq, dtype=torch.bool))], dim=-1).unsqueeze(0)
        p = p.masked_fill(~causal, float("-inf"))
        w = torch.exp(p - p.max(dim=-1)[0][:, :, None])
        gsum = w.sum(dim=-1)[:, :, None]
        w = (w * 256.0).to(torch.float8_e4m3fn).float()
        y = torch.matmul(w, v_batch) / gsum * (v_scale / 256.0)
        outs.append(y.transpose(0, 1).to(torch.bfloat16))
    return torch.stack(outs, 0)


def ref_prefill_fp8(q8, kcache8, vcache8, qscale, kscale, vscale, cu_seqlens_q, block_ids, seqlens_kv,
                    k_per_token=False, block_mask=None):
    """FP8 paged causal prefill oracle: restates naive_attn_with_kvcache_func of reference
    tests/test_attention_with_kvcache_qpertoken_perhead_kvpertensor_prefill_fp8.py:14-83 (per-tensor K/V
    scales; P quantised as e4m3(256 p) against the row maximum, O / sum * vscale / 256), generalised to a
    different q length per request.  q8 [total_q, Hq, D] e4m3, caches [nblk, P, Hkv, D] e4m3,
    qscale [B, Hq, pad], seqlens_kv = cached tokens including the q tokens.  Returns bf16 [total_q, Hq, D].
    k_per_token: kscale is the byte view of the cache tail rows, vscale [Hkv] - the scale handling of
    tests/test_attention_with_kvcache_qkpertoken_perhead_vperhead_prefill_fp8.py, same as ref_attn_fp8."""
    total_q, hq, d = q8.shape
    P, hkv = kcache8.shape[1], kcache8.shape[2]
    group = hq // hkv
    out = torch.empty(total_q, hq, vcache8.shape[3], dtype=torch.bfloat16)
    for b in range(cu_seqlens_q.numel() - 1):
        a0, a1 = int(cu_seqlens_q[b]), int(cu_seqlens_q[b + 1])
        sq, L = a1 - a0, int(seqlens_kv[b])
        if sq == 0:
            continue
        nblk = (L + P - 1) // P
        ids = block_ids[b, :nblk].long()
        BQ = q8[a0:a1].float().transpose(0, 1)  # [Hq, sq, D]
        BK = kcache8[ids].float().reshape(-1, hkv, d).transpose(0, 1)[:, :L].repeat_interleave(group, dim=0)
        BV = vcache8[ids].float().reshape(-1, hkv, vcache8.shape[3]).transpose(0, 1)[:, :L].repeat_interleave(group, dim=0)
        scale = qscale[b, :, :sq].unsqueeze(-1)
        scores = torch.matmul(BQ, BK.transpose(-2, -1)) * scale / math.sqrt(d)
        if k_per_token:
            ksb = (kscale[ids].contiguous().view(torch.float32).permute(0, 1, 3, 2).reshape(-1, hkv)
                   .transpose(0, 1)[:, :L].repeat_interleave(group, dim=0)).float()
            scores = scores * ksb.unsqueeze(1)
        else:
            scores = scores * kscale[0]
        if block_mask is not None:  # tests/test_attention_blocksparse_..._fp8.py:83-88: 128 x 128 tiles, rows = q index
            em = block_mask[b].bool().repeat_interleave(128, dim=-2)[:, :sq, :]
            em = em.repeat_interleave(128, dim=-1)[:, :, :L]
            scores = scores.masked_fill(~em, float("-inf"))
        mask = torch.tril(torch.ones(L, L, dtype=torch.bool))[L - sq :, :]
        scores = scores.masked_fill(~mask, float("-inf"))
        w = torch.exp(scores - scores.max(dim=-1, keepdim=True)[0])
        gsum = w.sum(dim=-1, keepdim=True)
        w8 = (w * 256.0).to(torch.float8_e4m3fn).float()
        o = torch.matmul(w8, BV) / gsum
        if k_per_token:
            o = o * vscale[:, None, None].repeat_interleave(group, dim=0) / 256.0
        else:
            o = o * (vscale[0] / 256.0)
        out[a0:a1] = o.transpose(0, 1).to(torch.bfloat16)
    return out


def ref_prefill_bf16(q, kcache, vcache, cu_seqlens_q, block_ids, seqlens_kv):
    """bf16 causal prefill oracle: naive_attn_with_kvcache_func of reference
    tests/test_attention_with_kvcache_prefill_bf16.py:19-63 (fp32 softmax over bf16 inputs), per-request
    lengths.  block_ids None: contiguous K/V [total, Hkv, D] sharing cu_seqlens_q
    (tests/test_attention_prefill_bf16.py:54-110, causal)."""
    total_q, hq, d = q.shape
    hkv = kcache.shape[-2]
    group = hq // hkv
    out = torch.empty(total_q, hq, vcache.shape[-1], dtype=torch.bfloat16)
    for b in range(cu_seqlens_q.numel() - 1):
        a0, a1 = int(cu_seqlens_q[b]), int(cu_seqlens_q[b + 1])
        sq = a1 - a0
        if sq == 0:
            continue
        if block_ids is None:
            L = sq
            K, V = kcache[a0:a1], vcache[a0:a1]
        else:
            L, P = int(seqlens_kv[b]), kcache.shape[1]
            ids = block_ids[b, : (L + P - 1) // P].long()
            K, V = kcache[ids].reshape(-1, hkv, d)[:L], vcache[ids].reshape(-1, hkv, vcache.shape[-1])[:L]
        BQ = q[a0:a1].float().transpose(0, 1)
        BK = K.float().transpose(0, 1).repeat_interleave(group, dim=0)
        BV = V.float().transpose(0, 1).repeat_interleave(group, dim=0)
        scores = torch.matmul(BQ, BK.transpose(-2, -1)) / math.sqrt(d)
        mask = torch.tril(torch.ones(L, L, dtype=torch.bool))[L - sq :, :]
        scores = scores.masked_fill(~mask, float("-inf"))
        out[a0:a1] = torch.matmul(F.softmax(scores, dim=-1), BV).transpose(0, 1).to(torch.bfloat16)
    return out
