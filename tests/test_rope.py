"""rope_norm_store_kv / rope_norm_store_kv_fp8 parity (grid of reference tests/test_rope.py:228-367):
HIP kernel vs the CPU oracle (oracle/rope.py), prefill and decode (mtp 0/1, align-8 padded batch)."""
import pytest
import torch

from oracle import rope as orc
from utils import allclose

F8 = torch.float8_e4m3fn


def kv_block_indices(num_blocks, block_size, seqlens, gen):
    per = [(s + block_size - 1) // block_size for s in seqlens]
    perm = torch.randperm(num_blocks, generator=gen)[: sum(per)].to(torch.int32)
    out = torch.zeros(len(seqlens), max(per), dtype=torch.int32)
    o = 0
    for i, n in enumerate(per):
        out[i, :n] = perm[o : o + n]
        o += n
    return out


def make_inputs(num_req, is_prefill, mtp, hq, hkv, seed, blk=64, nblocks=256, max_pos=2048):
    g = torch.Generator().manual_seed(seed)
    d = 128
    hidden = (hq + 2 * hkv) * d
    cos_sin = orc.generate_cos_sin_cache(max_pos, d)
    kcache = torch.randn(nblocks, blk, hkv, d, generator=g).bfloat16()
    vcache = torch.randn(nblocks, blk, hkv, d, generator=g).bfloat16()
    qw, kw = torch.randn(d, generator=g), torch.randn(d, generator=g)
    if is_prefill:
        req_len = torch.randint(20, 200, (num_req,), generator=g)
        q_len = torch.minimum((torch.rand(num_req, generator=g) * req_len).long() + 1, req_len)
        qkv = torch.randn(int(q_len.sum()), hidden, generator=g).bfloat16()
        q_index = torch.cat([torch.zeros(1, dtype=torch.long), q_len.cumsum(0)]).int()
        num_seqlen = req_len.int()
        kv_idx = kv_block_indices(nblocks, blk, req_len.tolist(), g)
        real = None
    else:
        tpr = mtp + 1
        upd = (torch.randint(20, 200, (num_req,), generator=g) + tpr)
        nr, pr, pb = num_req * tpr, (num_req * tpr + 7) // 8 * 8, (num_req + 7) // 8 * 8
        qkv = torch.zeros(pr, hidden).bfloat16()
        qkv[:nr] = torch.randn(nr, hidden, generator=g).bfloat16()
        q_index = torch.full((pb + 1,), pr, dtype=torch.int32)
        q_index[: num_req + 1] = torch.arange(0, (num_req + 1) * tpr, tpr, dtype=torch.int32)
        num_seqlen = torch.zeros(pb, dtype=torch.int32)
        num_seqlen[:num_req] = upd.int()
        ki = kv_block_indices(nblocks, blk, upd.tolist(), g)
        kv_idx = torch.zeros(pb, ki.shape[1], dtype=torch.int32)
        kv_idx[:num_req] = ki
        real = nr
    return qkv, num_seqlen, q_index, kcache, vcache, kv_idx, qw, kw, cos_sin, real


def oracle(inp, num_req, policy):
    qkv, ns, qi, kc, vc, ki, qw, kw, cs, real = inp
    kr, vr = kc.clone(), vc.clone()
    if real is not None:
        q = orc.rope_norm_ref(kr, vr, qkv[:real], cs, ns[:num_req], qi[: num_req + 1], ki[:num_req], qw, kw, policy)
    else:
        q = orc.rope_norm_ref(kr, vr, qkv, cs, ns, qi, ki, qw, kw, policy)
    return q, kr, vr


def written_mask(inp, num_req):
    """[blocks, P, Hkv, D] bool: cache cells the op must write (new tokens + zeroed tail of the last page)."""
    _, ns, qi, kc, _, ki = inp[:6]
    blk = kc.shape[1]
    m = torch.zeros(kc.shape, dtype=torch.bool)
    for r in range(num_req):
        sl, ql = int(ns[r]), int(qi[r + 1] - qi[r])
        for pos in range(sl - ql, sl):
            m[int(ki[r, pos // blk]), pos % blk] = True
        if ql > 0:
            m[int(ki[r, (sl - 1) // blk]), (sl - 1) % blk + 1 :] = True
    return m


def test_oracle_rope_properties():
    """CPU: rotation preserves the pair norms; policy 0 leaves V untouched and writes every new token."""
    inp = make_inputs(3, True, None, 8, 1, seed=1)
    q, kr, vr = oracle(inp, 3, 0)
    qkv, ns, qi, kc, vc, ki = inp[:6]
    qin = qkv[:, : 8 * 128].float().view(-1, 8, 128)
    n_in = qin[..., :64] ** 2 + qin[..., 64:] ** 2
    n_out = q.float()[..., :64] ** 2 + q.float()[..., 64:] ** 2
    assert torch.allclose(n_in, n_out, atol=0.15, rtol=2e-2)
    pos = int(ns[0]) - 1
    assert torch.equal(vr[int(ki[0, pos // 64]), pos % 64, 0], qkv[int(qi[1]) - 1, 9 * 128 :])


@pytest.mark.gpu
@pytest.mark.parametrize("hq,hkv", [(8, 1), (64, 8)])
@pytest.mark.parametrize("policy", [0, 1, 2])
@pytest.mark.parametrize("num_req", [7, 16])
@pytest.mark.parametrize("is_prefill,mtp", [(True, None), (False, 0), (False, 1)])
def test_rope_norm_store_kv(hq, hkv, policy, num_req, is_prefill, mtp):
    import hpc

    inp = make_inputs(num_req, is_prefill, mtp, hq, hkv, seed=num_req * 7 + policy)
    ref_q, kr, vr = oracle(inp, num_req, policy)
    qkv, ns, qi, kc, vc, ki, qw, kw, cs, real = [t.cuda() if torch.is_tensor(t) else t for t in inp]
    out_q = hpc.rope_norm_store_kv(kc, vc, qkv, cs, ns, qi, ki, is_prefill,
                                   q_norm_weight=qw if policy else None, k_norm_weight=kw if policy else None,
                                   qk_norm_policy=policy)
    rows = real if real is not None else int(qi[-1])
    assert allclose(ref_q, out_q[:rows].cpu(), atol=8e-2)
    assert allclose(kr, kc.cpu(), atol=8e-2)
    assert allclose(vr, vc.cpu(), atol=8e-2)
    # V is a pure copy and untouched pages stay bit-identical
    assert torch.equal(vr.view(torch.int16), vc.cpu().view(torch.int16))


@pytest.mark.gpu
def test_rope_bypass_outputs():
    """out_k / out_v given: caches stay untouched, K/V land in [rows, Hkv, 128] (reference hpc/rope.py:60-80)."""
    import hpc

    inp = make_inputs(5, True, None, 8, 1, seed=11)
    ref_q, kr, vr = oracle(inp, 5, 1)
    qkv, ns, qi, kc, vc, ki, qw, kw, cs, _ = [t.cuda() if torch.is_tensor(t) else t for t in inp]
    k0, v0 = kc.clone(), vc.clone()
    rows = qkv.shape[0]
    ok = torch.empty(rows, 1, 128, dtype=torch.bfloat16, device="cuda")
    ov = torch.empty_like(ok)
    oq = torch.empty(rows, 8, 128, dtype=torch.bfloat16, device="cuda")
    r = hpc.rope_norm_store_kv(kc, vc, qkv, cs, ns, qi, ki, True, qw, kw, oq, ok, ov, 1)
    assert r.data_ptr() == oq.data_ptr()
    assert torch.equal(kc, k0) and torch.equal(vc, v0)
    assert allclose(ref_q, oq.cpu(), atol=8e-2)
    assert torch.equal(ov.cpu().view(-1), qkv[:, 9 * 128 :].cpu().reshape(-1))
    # K rows: compare with what the oracle wrote into its cache at the token positions
    for req in range(5):
        sl, a, b = int(ns[req]), int(qi[req]), int(qi[req + 1])
        for t in range(a, b):
            pos = sl - (b - t)
            assert allclose(kr[int(ki[req, pos // 64]), pos % 64], ok[t].cpu(), atol=8e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("hq,hkv", [(8, 1), (64, 8)])
@pytest.mark.parametrize("policy", [0, 1, 2])
@pytest.mark.parametrize("quant_policy", [1, 2])
@pytest.mark.parametrize("num_req", [7, 16])
@pytest.mark.parametrize("is_prefill,mtp", [(True, None), (False, 0), (False, 1)])
def test_rope_norm_store_kv_fp8(hq, hkv, policy, quant_policy, num_req, is_prefill, mtp):
    import hpc

    if num_req == 16 and (policy != 1 or hq != 64):
        pytest.skip("the 16-request batch (no align-8 padding) is sampled once per mode / quant policy")

    inp = make_inputs(num_req, is_prefill, mtp, hq, hkv, seed=num_req * 5 + policy + 3 * quant_policy)
    ref_q, kr, vr = oracle(inp, num_req, policy)
    qkv, ns, qi, kc, vc, ki, qw, kw, cs, real = [t.cuda() if torch.is_tensor(t) else t for t in inp]
    k_scale = torch.tensor([0.1], device="cuda")
    v_scale = torch.tensor([0.1], device="cuda")
    q_scale_val = 2.0
    q_scale_inv = torch.tensor([1.0 / q_scale_val], device="cuda")
    kc8, vc8 = kc.to(F8), vc.to(F8)
    max_seqlens = int((qi[1:] - qi[:-1]).max()) if is_prefill else mtp + 1
    q8, q_scale, flag = hpc.rope_norm_store_kv_fp8(
        kc8, vc8, qkv, cs, ns, qi, ki, is_prefill, k_scale, v_scale, quant_policy, max_seqlens,
        q_scale_inv=q_scale_inv if quant_policy == 2 else None,
        q_norm_weight=qw if policy else None, k_norm_weight=kw if policy else None, qk_norm_policy=policy)
    assert flag.shape == (ns.shape[0], hkv) and flag.dtype == torch.int32
    assert torch.all(flag[:num_req] == 0)
    if quant_policy == 1:
        if is_prefill:
            pad = (max_seqlens + 127) // 128 * 128
            assert q_scale.shape == (ns.shape[0], hq, pad)
            lens = qi[1:] - qi[:-1]
            mask = torch.arange(pad, device="cuda").expand(ns.shape[0], pad) < lens.unsqueeze(1)
            flat = q_scale.permute(0, 2, 1)[mask]
            rows = int(qi[-1])
            q_bf16 = (q8[:rows].to(torch.bfloat16) * flat[:, :, None]).to(torch.bfloat16)
        else:
            assert q_scale.shape == (qkv.shape[0], hq)
            rows = real
            q_bf16 = (q8[:rows].to(torch.bfloat16) * q_scale[:rows, :, None]).to(torch.bfloat16)
            # dynamic scale: amax maps to the e4m3 maximum
            assert float(q8[:rows].float().abs().amax(-1).min()) >= 416.0
    else:
        assert q_scale is None
        rows = real if real is not None else q8.shape[0]
        q_bf16 = (q8[:rows].float() * q_scale_val).to(torch.bfloat16)
    assert allclose(ref_q, q_bf16.cpu(), atol=0.5)
    # caches: new tokens quantised with 1/k_scale, tails zeroed, untouched pages keep their bytes
    touched = torch.zeros(kc.shape[0], dtype=torch.bool)
    touched[ki[:num_req].cpu().long().reshape(-1)] = True
    kd, vd = kc8.float().cpu() * 0.1, vc8.float().cpu() * 0.1
    kq0, vq0 = inp[3].to(F8).float(), inp[4].to(F8).float()
    changed_k = written_mask(inp, num_req)
    assert torch.allclose(kd[changed_k], kr.float()[changed_k].clamp(-44.8, 44.8), atol=0.26, rtol=0.07)
    changed_v = changed_k
    assert torch.allclose(vd[changed_v], vr.float()[changed_v].clamp(-44.8, 44.8), atol=0.26, rtol=0.07)
    assert torch.equal(kc8.cpu()[~touched].view(torch.uint8), inp[3].to(F8)[~touched].view(torch.uint8))
    same = ~changed_k
    assert torch.equal(kc8.float().cpu()[same], kq0[same])
    assert torch.equal(vc8.float().cpu()[~changed_v], vq0[~changed_v])


@pytest.mark.gpu
def test_rope_fp8_upper_max_checked():
    import hpc

    inp = make_inputs(2, False, 0, 8, 1, seed=3)
    qkv, ns, qi, kc, vc, ki, qw, kw, cs, real = [t.cuda() if torch.is_tensor(t) else t for t in inp]
    one = torch.ones(1, device="cuda")
    with pytest.raises(RuntimeError):
        hpc.rope_norm_store_kv_fp8(kc.to(F8), vc.to(F8), qkv, cs, ns, qi, ki, False, one, one, 1, 1, upper_max=500.0)
    q8, sc, _ = hpc.rope_norm_store_kv_fp8(kc.to(F8), vc.to(F8), qkv, cs, ns, qi, ki, False, one, one, 1, 1,
                                           upper_max=224.0)
    assert float(q8[:real].float().abs().max()) <= 224.0
