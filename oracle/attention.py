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
