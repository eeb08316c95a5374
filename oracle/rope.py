"""RoPE + QK-norm + paged KV store oracle — TEST INFRASTRUCTURE (see oracle/__init__.py).
Restates reference tests/test_rope.py:14-117 (generate_cos_sin_cache, apply_rms_norm_reference,
apply_rotary_pos_emb_neox_reference, rope_norm_ref) for CPU tensors."""
import torch


def generate_cos_sin_cache(max_position, head_dim, base=10000.0):
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    freqs = torch.outer(torch.arange(max_position).float(), inv_freq)
    return torch.cat([freqs.cos(), freqs.sin()], dim=-1)


def rms_norm(x, weight, eps=1e-6):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * weight


def rotary_neox(x, cos_sin):
    h = x.shape[-1] // 2
    x1, x2 = x[..., :h], x[..., h:]
    c, s = cos_sin[:, :h].unsqueeze(1), cos_sin[:, h:].unsqueeze(1)
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1)


def rope_norm_ref(kcache, vcache, qkv, cos_sin, num_seqlen_per_req, q_index, kv_indices, q_norm_weight,
                  k_norm_weight, qk_norm_policy):
    """Updates kcache / vcache in place, returns q (reference tests/test_rope.py:47-117)."""
    dtype = qkv.dtype
    num_kv, v_dim, qk_dim, blk = kcache.shape[2], vcache.shape[3], kcache.shape[3], kcache.shape[1]
    num_q = (qkv.shape[1] - num_kv * qk_dim - num_kv * v_dim) // qk_dim
    num_req = num_seqlen_per_req.shape[0]
    q_lens = (q_index[1:] - q_index[:-1]).tolist()
    num_rows = int(q_index[-1])
    q = qkv[:, : num_q * qk_dim].float().view(num_rows, num_q, qk_dim)
    k = qkv[:, num_q * qk_dim : (num_q + num_kv) * qk_dim].float().view(num_rows, num_kv, qk_dim)
    v = qkv[:, (num_q + num_kv) * qk_dim :].view(num_rows, num_kv, v_dim)
    cs = torch.zeros(num_rows, qk_dim, dtype=torch.float32)
    off = 0
    for i in range(num_req):
        sl, ql = int(num_seqlen_per_req[i]), q_lens[i]
        if ql > 0:
            cs[off : off + ql] = cos_sin[sl - ql : sl]
        off += ql
    if qk_norm_policy == 2:
        q, k = rms_norm(q, q_norm_weight), rms_norm(k, k_norm_weight)
    q, k = rotary_neox(q, cs), rotary_neox(k, cs)
    if qk_norm_policy == 1:
        q, k = rms_norm(q, q_norm_weight), rms_norm(k, k_norm_weight)
    tok = 0
    for ri in range(num_req):
        sl, ql = int(num_seqlen_per_req[ri]), q_lens[ri]
        for pos in range(sl - ql, sl):
            bi, pb = pos // blk, pos % blk
            cb = int(kv_indices[ri, bi])
            kcache[cb, pb] = k[tok].to(dtype)
            vcache[cb, pb] = v[tok].to(dtype)
            if pos == sl - 1 and pb + 1 < blk:
                kcache[cb, pb + 1 :] = 0
                vcache[cb, pb + 1 :] = 0
            tok += 1
    return q.to(dtype)
