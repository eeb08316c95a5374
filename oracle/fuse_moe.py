"""Fused-MoE (FP8 blockwise) oracles — TEST INFRASTRUCTURE (see oracle/__init__.py).

CPU restatements of the PyTorch-eager references of reference tests/test_fuse_moe_blockwise.py:
  gather_expert_inputs              :23-78   (stable slotting in flattened (token, k) order)
  group_gemm_blockwise              :81-138  (per 128x128 block: fp32 dot * xs * ws, bf16 round)
  act_mul_and_blockwise_quant       :141-199 (x/(1+exp(-x)) * up; scale = amax/448; 1/(scale+1e-8))
  reduce                            :202-217
  fuse_moe_blockwise_fp8            :220-262
The per-block GEMM uses an fp32 matmul on the upcast e4m3 operands instead of torch._scaled_mm
with unit scales (same products - e4m3 x e4m3 is exact in fp32 - different summation order).
"""
import torch


def gather_expert_inputs(x, x_scale, topk_ids, num_expert, rank_ep):
    num_tokens, num_topk = topk_ids.shape
    start, end = rank_ep * num_expert, (rank_ep + 1) * num_expert
    flat = topk_ids.flatten()
    counts = torch.zeros(num_expert, dtype=torch.int32)
    for e in range(num_expert):
        counts[e] = int((flat == start + e).sum())
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(counts, 0).to(torch.int32)])
    y = torch.zeros((num_tokens * num_topk, x.shape[1]), dtype=x.dtype)
    y_scale = torch.zeros((num_tokens * num_topk, x_scale.size(1)), dtype=torch.float32)
    token_pos = torch.full((num_tokens, num_topk), -1, dtype=torch.int32)
    fill = torch.zeros(num_expert, dtype=torch.int64)
    yb, xb = y.view(torch.uint8), x.view(torch.uint8)
    for idx, ie in enumerate(flat.tolist()):
        if start <= ie < end:
            pos = int(cu[ie - start]) + int(fill[ie - start])
            yb[pos] = xb[idx // num_topk]
            y_scale[pos] = x_scale[idx // num_topk]
            token_pos[idx // num_topk, idx % num_topk] = pos
            fill[ie - start] += 1
    return y, y_scale, token_pos, counts, cu


def group_gemm_blockwise(x, w, seqlens, cu_seqlens, xscale, wscale):
    """x e4m3 [M, K]; w e4m3 [G, N, K]; xscale f32 [M, K/128] (row-major); wscale [G, N/128, >=K/128]."""
    m, k = x.shape
    num_group, n, _ = w.shape
    kb = k // 128
    y = torch.zeros((m, n), dtype=torch.bfloat16)
    for i in range(num_group):
        s, cnt = int(cu_seqlens[i]), int(seqlens[i])
        if cnt == 0:
            continue
        xg = x[s : s + cnt].float().reshape(cnt, kb, 128)
        wg = w[i].float().reshape(n, kb, 128)
        part = torch.einsum("mbk,nbk->mnb", xg, wg)                     # [cnt, n, kb]
        ws = wscale[i][:, :kb].repeat_interleave(128, dim=0)            # [n, kb]
        out = torch.zeros((cnt, n), dtype=torch.float32)
        for b in range(kb):                                             # same += order as the reference
            out += part[:, :, b] * xscale[s : s + cnt, b].unsqueeze(1) * ws[:, b].unsqueeze(0)
        y[s : s + cnt] = out.to(torch.bfloat16)
    return y


def group_gemm_blockwise_kernel_arith(x, w, seqlens, cu_seqlens, xscale, wscale):
    """The same GEMM in the arithmetic of the reference KERNEL instead of the reference test's eager model
    (src/group_gemm/kernels.cuh:808-834): per 128-wide k block the fp32 partial of the block (WGMMA from a zeroed
    accumulator), `yscale = xs[m, kb] * ws[n / 128, kb]` (one fp32 multiply), then `tDr = tCr * yscale + tDr` - a
    fused multiply-add, ONE rounding per k block - where the eager model (`group_gemm_blockwise` above =
    tests/test_fuse_moe_blockwise.py:115-128) rounds `(part * xs) * ws` twice and adds with a third rounding.  The
    two differ in the last fp32 bit of some sums, which moves a bf16 output by one ulp when the sum sits on a bf16
    tie (and, in the fused MoE, one e4m3 code of the quantised activation when that bf16 value sits on an e4m3 tie).
    The FMA is emulated in float64: the 48-bit product is exact, the sum is rounded once to 53 bits and once more to
    fp32 - a double rounding that differs from a true FMA only when the 53-bit sum lands exactly on an fp32 tie (odds
    2^-29 per operation, and visible in bf16 only on a further 2^-8 coincidence).  HIP's blockwise GEMMs use this
    arithmetic; what is left between them and this function is the matrix pipe's rounding of the 128-term block sums
    (here: exact sums rounded once to fp32), which moves ~1e-4 of the bf16 outputs by one ulp
    (tests/test_fuse_moe_blockwise.py::test_group_gemm_blockwise_is_the_reference_kernel_arithmetic)."""
    m, k = x.shape
    num_group, n, _ = w.shape
    kb = k // 128
    y = torch.zeros((m, n), dtype=torch.bfloat16)
    for i in range(num_group):
        s, cnt = int(cu_seqlens[i]), int(seqlens[i])
        if cnt == 0:
            continue
        xg = x[s : s + cnt].to(torch.float64).reshape(cnt, kb, 128)
        wg = w[i].to(torch.float64).reshape(n, kb, 128)
        part = torch.einsum("mbk,nbk->mnb", xg, wg).float()                     # the block partials, rounded once
        ws = wscale[i][:, :kb].repeat_interleave(128, dim=0)                     # [n, kb]
        out = torch.zeros((cnt, n), dtype=torch.float32)
        for b in range(kb):
            ys = xscale[s : s + cnt, b].unsqueeze(1) * ws[:, b].unsqueeze(0)   # fp32 multiply (kernels.cuh:811)
            out = (part[:, :, b].double() * ys.double() + out.double()).float()  # FFMA (kernels.cuh:834)
        y[s : s + cnt] = out.to(torch.bfloat16)
    return y


def act_mul_and_blockwise_quant(gate_up_out):
    gate, up = torch.chunk(gate_up_out.float(), 2, dim=1)
    out = gate / (1 + (-gate).exp()) * up
    rows, feats = out.shape
    nblk = (feats + 127) // 128
    q = torch.empty(rows, feats, dtype=torch.float8_e4m3fn)
    scales = torch.empty(rows, nblk, dtype=torch.float32)
    for b in range(nblk):
        blk = out[:, b * 128 : (b + 1) * 128]
        scale = blk.abs().amax(dim=1, keepdim=True) / 448.0
        inv = 1.0 / (scale + 1e-8)
        scales[:, b] = scale.squeeze(1)
        q[:, b * 128 : (b + 1) * 128] = (blk * inv).to(torch.float8_e4m3fn)
    return q, scales


def reduce(x_bf16, topk_pos, topk_scale, shared_output=None):
    num_tokens, num_topk = topk_pos.shape
    pos = topk_pos.long().clamp_min(0)
    rows = x_bf16.float()[pos]                                           # [T, k, H]
    w = (topk_scale.float() * (topk_pos >= 0)).unsqueeze(-1)
    acc = torch.zeros(num_tokens, x_bf16.shape[1], dtype=torch.float32)
    for j in range(num_topk):                                            # same accumulation order
        acc += rows[:, j] * w[:, j]
    if shared_output is not None:
        acc += shared_output.float()
    return acc.to(torch.bfloat16)


def fuse_moe_blockwise_fp8(x, x_scale, gate_up_weight, gate_up_weight_scale, down_weight,
                           down_weight_scale, topk_ids, topk_scale, rank_ep, num_expert,
                           shared_output=None, return_intermediates=False, kernel_arith=False):
    """kernel_arith: both GEMMs in the reference kernel's arithmetic (group_gemm_blockwise_kernel_arith) instead of
    the reference test's eager model - used to tell what a literal-bar miss of an implementation is made of."""
    num_expert_local = gate_up_weight.size(0)
    gi, gis, topk_pos, seqlens, cu = gather_expert_inputs(x, x_scale, topk_ids, num_expert_local, rank_ep)
    gemm = group_gemm_blockwise_kernel_arith if kernel_arith else group_gemm_blockwise
    g = gemm(gi, gate_up_weight, seqlens, cu, gis, gate_up_weight_scale)
    di, dis = act_mul_and_blockwise_quant(g)
    d = gemm(di, down_weight, seqlens, cu, dis, down_weight_scale)
    y = reduce(d, topk_pos, topk_scale, shared_output)
    if return_intermediates:
        return y, dict(topk_pos=topk_pos, seqlens=seqlens, cu_seqlens=cu, gate_up=g, down_in=di,
                       down_in_scale=dis, down_out=d)
    return y


# ---- per-tensor path: reference tests/test_fuse_moe_pertensor.py:73-160, test_fuse_moe_cp_async.py:65-143,
#      test_act.py:20-29, test_group_gemm_pertensor.py:20-48 ------------------------------------------------
def group_gemm_pertensor(x, w, seqlens, cu_seqlens, scale):
    m, _ = x.shape
    num_group, n, _ = w.shape
    y = torch.zeros((m, n), dtype=torch.bfloat16)
    for i in range(num_group):
        s, cnt = int(cu_seqlens[i]), int(seqlens[i])
        if cnt == 0:
            continue
        y[s : s + cnt] = ((x[s : s + cnt].float() @ w[i].float().t()) * scale[i].float()).to(torch.bfloat16)
    return y


def act_mul_and_quant(gate_up, scale, use_bf16_mul=True):
    gate, up = torch.chunk(gate_up.float(), 2, dim=1)
    silu = gate / (1 + (-gate).exp())
    if use_bf16_mul:
        out = (silu.to(torch.bfloat16) * up.to(torch.bfloat16)).float() * scale
    else:
        out = silu * up * scale
    return out.to(torch.float8_e4m3fn)


def scaled_fp8_quant(x, scale):
    """reference scaled_fp8_quant_kernel, src/activation/activation.cu:461-505 (entry src/activation/entry.cc:158-200):
    inv_scale = 1.0f / scale[0] once, every element of the fp32 / fp16 / bf16 input is MULTIPLIED by it in fp32 and
    converted to e4m3 with saturation (__nv_fp8_e4m3(float) is a satfinite conversion; torch's cast turns values
    beyond the format into NaN, so the clamp is explicit here); returns (output, scale) like the entry.
    Pinned in tests/golden/make_golden.py against the reference benchmark's eager form `scaled_fp8_quant_local`
    (benchmark/fused_moe/backends/base.py:64-67, `(x.float() / scale).to(fp8)`): bit-equal for power-of-two scales,
    and for a general scale equal except where x / s and x * (1 / s) round to different sides of an e4m3 tie."""
    inv = torch.ones((), dtype=torch.float32) / scale.detach().float().reshape(-1)[0]
    y = x.float() * inv
    y = torch.where(torch.isnan(y), y, y.clamp(-448.0, 448.0))
    return y.to(torch.float8_e4m3fn), scale


def fuse_moe_pertensor_fp8(x, gate_up_weight, down_weight, gate_up_scale, down_scale, act_and_mul_scale,
                           topk_ids, topk_scale, rank_ep, shared_output=None, use_bf16_mul=True):
    num_expert = gate_up_weight.size(0)
    dummy = torch.zeros(x.shape[0], 1)
    gi, _, topk_pos, seqlens, cu = gather_expert_inputs(x, dummy, topk_ids, num_expert, rank_ep)
    g = group_gemm_pertensor(gi, gate_up_weight, seqlens, cu, gate_up_scale)
    di = act_mul_and_quant(g, act_and_mul_scale, use_bf16_mul)
    d = group_gemm_pertensor(di, down_weight, seqlens, cu, down_scale)
    return reduce(d, topk_pos, topk_scale, shared_output)


def fuse_moe_blockwise_fp8_rows(x, x_scale, expert_weights, topk_ids, topk_scale, rows, rank_ep,
                                num_expert_local, shared_output=None):
    """fuse_moe_blockwise_fp8 restricted to the token rows `rows` (every stage of the pipeline is
    row-independent: GEMM rows, per-row 128-block quantisation, per-token reduce), with the expert
    weights streamed one expert at a time - the full-size configuration (64 experts x 135 MB of
    e4m3) never has to sit in host memory as fp32.  `expert_weights(e)` returns the CPU tensors
    (gate_up_weight[e], gate_up_weight_scale[e], down_weight[e], down_weight_scale[e]).
    Same arithmetic and rounding points as fuse_moe_blockwise_fp8 above (it calls the same stage
    functions); returns bf16 [len(rows), H]."""
    rows = [int(r) for r in rows]
    num_topk = topk_ids.shape[1]
    start = rank_ep * num_expert_local
    by_expert = {}
    for ri, t in enumerate(rows):
        for j in range(num_topk):
            e = int(topk_ids[t, j]) - start
            if 0 <= e < num_expert_local:
                by_expert.setdefault(e, []).append((ri, j))
    hidden = x.shape[1]
    down = torch.zeros(len(rows), num_topk, hidden, dtype=torch.bfloat16)
    valid = torch.zeros(len(rows), num_topk, dtype=torch.float32)
    for e in sorted(by_expert):
        pairs = by_expert[e]
        guw, guws, dw, dws = expert_weights(e)
        toks = torch.tensor([rows[ri] for ri, _ in pairs], dtype=torch.long)
        cnt = len(pairs)
        one, zero = torch.tensor([cnt], dtype=torch.int32), torch.tensor([0], dtype=torch.int32)
        g = group_gemm_blockwise(x[toks], guw[None], one, zero, x_scale[toks], guws[None])
        di, dis = act_mul_and_blockwise_quant(g)
        d = group_gemm_blockwise(di, dw[None], one, zero, dis, dws[None])
        for i, (ri, j) in enumerate(pairs):
            down[ri, j] = d[i]
            valid[ri, j] = 1.0
    tr = torch.tensor(rows, dtype=torch.long)
    acc = torch.zeros(len(rows), hidden, dtype=torch.float32)
    for j in range(num_topk):  # same accumulation order as reduce()
        acc += down[:, j].float() * (topk_scale[tr, j].float() * valid[:, j]).unsqueeze(-1)
    if shared_output is not None:
        acc += shared_output[tr].float()
    return acc.to(torch.bfloat16)
