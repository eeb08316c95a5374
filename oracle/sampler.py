"""Fused sampler oracle — TEST INFRASTRUCTURE (see oracle/__init__.py).
Restates reference tests/test_sampler.py:47-165 (ref_fused_sampler) and :430-436 / :490-505 (temperature
fast path) for CPU tensors.  One deliberate tightening: the reference model calls torch.topk, whose order
among EQUAL values is unspecified (and differs between devices); here the top-k is the stable descending
order - equal values rank by smaller token id - which is the kernel's rule and a valid torch.topk answer.
"""
import torch


def gumbel0_like(logits, generator=None):
    u = torch.rand(logits.shape, dtype=torch.float32, generator=generator).clamp_min_(1e-20)
    return -(-u.log()).log()


def stable_topk(row, k):
    vals, idx = torch.sort(row, descending=True, stable=True)
    return vals[:k], idx[:k]


def ref_fused_sampler(logits, *, penalty_mask=None, slot_id=None, repetition_penalty=0.0, temperature=0.0,
                      softmax_policy=0, topk=0, topp=0.0, max_topk=32, gumbel_noise):
    """Returns (token_ids [B,1] int32, expected penalty_mask after write-back or None)."""
    B, V = logits.shape
    work = logits.float().clone()

    def as_tensor(x, dtype):
        return x.to(dtype) if isinstance(x, torch.Tensor) else torch.full((B,), float(x), dtype=dtype)

    rp, temp, tp = (as_tensor(v, torch.float32) for v in (repetition_penalty, temperature, topp))
    tk = topk.to(torch.int64) if isinstance(topk, torch.Tensor) else torch.full((B,), int(topk), dtype=torch.int64)
    if penalty_mask is not None and slot_id is not None:
        for b in range(B):
            r = rp[b].item()
            if r <= 0:
                continue
            row = penalty_mask[int(slot_id[b])]
            bits = torch.zeros(row.numel() * 8, dtype=torch.bool)
            for bit in range(8):
                bits[bit::8] = ((row >> bit) & 1).bool()
            keep = bits[:V]
            wb = work[b]
            pos, neg = keep & (wb > 0), keep & (wb <= 0)
            wb[pos] = wb[pos] * (1.0 / r)
            wb[neg] = wb[neg] * r
    for b in range(B):
        t = temp[b].item()
        if t > 0:
            work[b] = work[b] / t
    if softmax_policy == 1:
        work = torch.softmax(work, dim=-1)
    tokens = torch.empty((B, 1), dtype=torch.int32)
    for b in range(B):
        k_b = int(tk[b])
        if k_b <= 0 or k_b > max_topk:
            k_b = max_topk
        vals, idx = stable_topk(work[b], k_b)
        if softmax_policy == 2:
            probs = torch.softmax(vals, dim=-1)
            g = probs.log()
        elif softmax_policy == 1:
            probs = vals
            g = torch.where(probs > 0, probs.log(), torch.full_like(probs, float("-inf")))
        else:
            probs, g = None, vals
        keep = torch.ones(k_b, dtype=torch.bool)
        if tp[b].item() > 0:
            cumsum = torch.cumsum(probs, dim=-1)
            keep = (torch.arange(k_b) == 0) | ((cumsum - probs) < tp[b].item())
        key = g + gumbel_noise[b, idx]
        key = torch.where(keep, key, torch.full_like(key, float("-inf")))
        cand = (key == key.max()).nonzero(as_tuple=True)[0]
        best = cand[torch.argmin(idx[cand])] if cand.numel() else torch.tensor(0)
        tokens[b, 0] = int(idx[best])
    expected = None
    if penalty_mask is not None and slot_id is not None:
        expected = penalty_mask.clone()
        for b in range(B):
            tok = int(tokens[b, 0])
            expected[int(slot_id[b]), tok // 8] |= 1 << (tok % 8)
    return tokens, expected


def ref_temperature_sample(logits, temperature, gumbel, draft_token_ids=None):
    scaled = logits.float() / temperature.view(-1, 1).float()
    if draft_token_ids is not None:
        V = logits.shape[1]
        valid = (draft_token_ids >= 0) & (draft_token_ids < V)
        if valid.any():
            rows = torch.nonzero(valid, as_tuple=False).squeeze(1)
            scaled[rows, draft_token_ids[valid]] = float("-inf")
    return (scaled + gumbel.float()).argmax(dim=-1).to(torch.int32).view(-1, 1)
