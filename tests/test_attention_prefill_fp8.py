"""attention_with_kvcache_prefill_fp8 parity (grid of reference
tests/test_attention_with_kvcache_qpertoken_perhead_kvpertensor_prefill_fp8.py:93-236, atol 0.1 there)."""
import math

import pytest
import torch

from oracle import attention as oattn
from utils import allclose

F8 = torch.float8_e4m3fn


def make_case(seq_q, seq_kv, hq, hkv, block_size, seed=10086):
    g = torch.Generator().manual_seed(seed)
    B, D = len(seq_q), 128
    total_q = sum(seq_q)
    q = (torch.randn(total_q, hq, D, generator=g) / math.sqrt(D)).bfloat16().to(F8)
    pad = (max(seq_q) + 127) // 128 * 128
    qscale = torch.randn(B, hq, pad, generator=g).abs() / 10
    kscale = torch.randn(1, generator=g).abs() * 10
    vscale = torch.randn(1, generator=g)
    lens = torch.tensor(seq_kv, dtype=torch.int32)
    nblk = (lens + block_size - 1) // block_size
    max_blocks = int(nblk.sum()) * 2
    kv = torch.randn(max_blocks, 2, block_size, hkv, D, generator=g).bfloat16().to(F8)
    perm = torch.randperm(max_blocks, generator=g)[: int(nblk.sum())].to(torch.int32)
    block_ids = torch.zeros(B, int(nblk.max()), dtype=torch.int32)
    o = 0
    for i in range(B):
        block_ids[i, : int(nblk[i])] = perm[o : o + int(nblk[i])]
        o += int(nblk[i])
    cu = torch.tensor([0] + list(torch.tensor(seq_q).cumsum(0)), dtype=torch.int32)
    return q, kv, qscale, kscale, vscale, cu, block_ids, lens


def test_oracle_prefill_reduces_to_decode_oracle():
    """CPU: with one q token per request the prefill model is the fp8 decode model (same P quantisation)."""
    q, kv, qscale, kscale, vscale, cu, bid, lens = make_case([1, 1, 1], [70, 200, 64], 8, 1, 64)
    out = oattn.ref_prefill_fp8(q, kv[:, 0], kv[:, 1], qscale, kscale, vscale, cu, bid, lens)
    nblocks = (lens + 63) // 64
    dec = oattn.ref_attn_fp8(q, kv, bid, nblocks, 1, lens - 1, qscale[:, :, 0], kscale, vscale, False)
    assert allclose(dec, out, atol=2e-2, rtol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("kv_layout", ["nhd", "hnd"])
@pytest.mark.parametrize("num_seq_q", [100, 500, 1000, 1500, 3904])
@pytest.mark.parametrize("use_output", [False, True])
def test_attention_with_kvcache_prefill_fp8(kv_layout, num_seq_q, use_output):
    import hpc

    if use_output and (kv_layout == "hnd" or num_seq_q not in (100, 1500)):
        pytest.skip("output= variant sampled")
    B, hq, hkv, seq_kv = 4, 4, 1, 3904
    q, kv, qscale, kscale, vscale, cu, bid, lens = make_case([num_seq_q] * B, [seq_kv] * B, hq, hkv, 64)
    gt = oattn.ref_prefill_fp8(q, kv[:, 0], kv[:, 1], qscale, kscale, vscale, cu, bid, lens)
    kvd = kv.cuda()
    kc, vc = kvd[:, 0], kvd[:, 1]
    if kv_layout == "hnd":
        kc = kc.view(torch.uint8).transpose(1, 2).contiguous().transpose(1, 2).view(F8)
        vc = vc.view(torch.uint8).transpose(1, 2).contiguous().transpose(1, 2).view(F8)
    out = torch.empty(q.shape, dtype=torch.bfloat16, device="cuda") if use_output else None
    my = hpc.attention_with_kvcache_prefill_fp8(q.cuda(), kc, vc, qscale.cuda(), kscale.cuda(), vscale.cuda(), cu.cuda(),
                                                bid.cuda(), lens.cuda(), num_seq_q, output=out)
    if use_output:
        assert my.data_ptr() == out.data_ptr()
    assert my.dtype == torch.bfloat16
    assert allclose(gt, my.cpu(), atol=0.1, rtol=0.02)


@pytest.mark.gpu
@pytest.mark.parametrize("hq,hkv", [(8, 1), (32, 4), (16, 8), (4, 4)])
@pytest.mark.parametrize("block_size", [16, 64])
def test_prefill_fp8_ragged_requests(hq, hkv, block_size):
    """different q / kv lengths per request, q == kv (no cached prefix), single-token requests, GQA groups."""
    import hpc

    seq_q = [1, 37, 128, 300, 5, 64]
    seq_kv = [1, 37, 500, 300, 1000, 65]
    q, kv, qscale, kscale, vscale, cu, bid, lens = make_case(seq_q, seq_kv, hq, hkv, block_size, seed=7)
    gt = oattn.ref_prefill_fp8(q, kv[:, 0], kv[:, 1], qscale, kscale, vscale, cu, bid, lens)
    kvd = kv.cuda()
    my = hpc.attention_with_kvcache_prefill_fp8(q.cuda(), kvd[:, 0], kvd[:, 1], qscale.cuda(), kscale.cuda(),
                                                vscale.cuda(), cu.cuda(), bid.cuda(), lens.cuda(), max(seq_q))
    assert allclose(gt, my.cpu(), atol=0.1, rtol=0.02)


@pytest.mark.gpu
def test_prefill_fp8_errors():
    import hpc

    q, kv, qscale, kscale, vscale, cu, bid, lens = make_case([8], [8], 4, 1, 64)
    d = [t.cuda() for t in (q, kv[:, 0], kv[:, 1], qscale, kscale, vscale, cu, bid, lens)]
    with pytest.raises(RuntimeError):
        hpc.attention_with_kvcache_prefill_fp8(d[0].to(torch.bfloat16), *d[1:], 8)
    with pytest.raises(RuntimeError):
        hpc.attention_with_kvcache_prefill_fp8(*d, 8, output=torch.empty(8, 4, 128, device="cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("hq,hkv", [(8, 1), (16, 4)])
@pytest.mark.parametrize("block_size", [32, 64])
def test_prefill_fp8_k_per_token(hq, hkv, block_size):
    """quant_type QPERTOKEN_PERHEAD_KPERTOKEN_PERHEAD_VPERHEAD: K scales in the page tail rows, V per head."""
    import hpc

    seq_q, seq_kv = [200, 1, 64, 333], [700, 90, 64, 333]
    q, kv, qscale, _, _, cu, bid, lens = make_case(seq_q, seq_kv, hq, hkv, block_size, seed=11)
    g = torch.Generator().manual_seed(5)
    rows = block_size * 4 // 128
    raw = torch.randn(kv.shape[0], 2, block_size + rows, hkv, 128, generator=g).bfloat16()
    kc, _ = oattn.quant_paged_cache_pertoken(raw[:, 0], block_size)
    vc, vscale = oattn.quant_paged_cache_perhead(raw[:, 1], block_size)
    kscale = kc[:, block_size:]
    gt = oattn.ref_prefill_fp8(q, kc[:, :block_size], vc[:, :block_size], qscale, kscale, vscale, cu, bid, lens,
                               k_per_token=True)
    kcd, vcd = kc.cuda(), vc.cuda()
    my = hpc.attention_with_kvcache_prefill_fp8(
        q.cuda(), kcd[:, :block_size], vcd[:, :block_size], qscale.cuda(), kcd[:, block_size:], vscale.cuda(),
        cu.cuda(), bid.cuda(), lens.cuda(), max(seq_q),
        quant_type=hpc.QuantType.QPERTOKEN_PERHEAD_KPERTOKEN_PERHEAD_VPERHEAD)
    assert allclose(gt, my.cpu(), atol=0.1, rtol=0.02)


def block_sparse_mask(batch, heads, nrow, ncol, skip_ratio, gen):
    """reference tests/test_attention_blocksparse_qpertoken_perhead_kvpertensor_fp8.py:21-35 (True = attend,
    causal upper part cleared, the diagonal tile of every q tile kept)."""
    mask = torch.rand(batch, heads, nrow, ncol, generator=gen) >= skip_ratio
    row = torch.arange(nrow).view(nrow, 1)
    col = torch.arange(ncol).view(1, ncol)
    boundary = row + (ncol - nrow)
    mask = mask & (col <= boundary)
    return mask | (col == torch.clamp(boundary, max=ncol - 1))


@pytest.mark.gpu
@pytest.mark.parametrize("kv_layout", ["nhd", "hnd"])
@pytest.mark.parametrize("num_seq", [1024, 2048])
@pytest.mark.parametrize("skip_ratio", [0.0, 0.5, 0.9])
@pytest.mark.parametrize("hq,hkv", [(4, 1), (16, 2), (32, 2)])  # (32, 2): 16 q heads per kv head - mask bits 8 ... 15
def test_blocksparse_prefill_fp8(kv_layout, num_seq, skip_ratio, hq, hkv):
    import hpc

    B = 2
    q, kv, qscale, kscale, vscale, cu, bid, lens = make_case([num_seq] * B, [num_seq] * B, hq, hkv, 64, seed=21)
    g = torch.Generator().manual_seed(4)
    nt = (num_seq + 127) // 128
    bm = block_sparse_mask(B, hq, nt, nt, skip_ratio, g) if skip_ratio > 0 else None
    gt = oattn.ref_prefill_fp8(q, kv[:, 0], kv[:, 1], qscale, kscale, vscale, cu, bid, lens, block_mask=bm)
    kvd = kv.cuda()
    kc, vc = kvd[:, 0], kvd[:, 1]
    if kv_layout == "hnd":
        kc = kc.view(torch.uint8).transpose(1, 2).contiguous().transpose(1, 2).view(F8)
        vc = vc.view(torch.uint8).transpose(1, 2).contiguous().transpose(1, 2).view(F8)
    my = hpc.attention_with_kvcache_blocksparse_prefill_fp8(
        q.cuda(), kc, vc, qscale.cuda(), kscale.cuda(), vscale.cuda(), cu.cuda(), bid.cuda(), lens.cuda(), num_seq,
        block_mask=None if bm is None else bm.to(torch.uint8).cuda())
    assert allclose(gt, my.cpu(), atol=0.1, rtol=0.02)


@pytest.mark.gpu
def test_blocksparse_prefill_fp8_cached_prefix_and_ragged():
    """q shorter than kv (mask rows index q positions, columns absolute kv tiles) and ragged requests."""
    import hpc

    seq_q, seq_kv = [300, 129, 1], [900, 129, 640]
    hq, hkv = 8, 2
    q, kv, qscale, kscale, vscale, cu, bid, lens = make_case(seq_q, seq_kv, hq, hkv, 32, seed=8)
    g = torch.Generator().manual_seed(6)
    nrow, ncol = (max(seq_q) + 127) // 128, (max(seq_kv) + 127) // 128
    bm = torch.rand(len(seq_q), hq, nrow, ncol, generator=g) >= 0.4
    for b, (sq, L) in enumerate(zip(seq_q, seq_kv)):  # keep every row's own diagonal tile
        for r in range(nrow):
            last_pos = min(sq - 1, r * 128 + 127)
            if r * 128 < sq:
                bm[b, :, r, (L - sq + last_pos) // 128] = True
                bm[b, :, r, (L - sq + r * 128) // 128] = True
    gt = oattn.ref_prefill_fp8(q, kv[:, 0], kv[:, 1], qscale, kscale, vscale, cu, bid, lens, block_mask=bm)
    kvd = kv.cuda()
    my = hpc.attention_with_kvcache_blocksparse_prefill_fp8(
        q.cuda(), kvd[:, 0], kvd[:, 1], qscale.cuda(), kscale.cuda(), vscale.cuda(), cu.cuda(), bid.cuda(),
        lens.cuda(), max(seq_q), block_mask=bm.to(torch.uint8).cuda())
    assert allclose(gt, my.cpu(), atol=0.1, rtol=0.02)
