"""bf16 prefill attention parity: attention_with_kvcache_prefill_bf16 (grid of reference
tests/test_attention_with_kvcache_prefill_bf16.py:78-160) and attention_prefill_bf16
(tests/test_attention_prefill_bf16.py:124-190); the reference compares at atol 0.016 / rtol 0.016-ish on bf16."""
import math

import pytest
import torch

from oracle import attention as oattn
from utils import allclose


def paged_case(seq_q, seq_kv, hq, hkv, block_size, seed=41):
    g = torch.Generator().manual_seed(seed)
    D = 128
    q = (torch.randn(sum(seq_q), hq, D, generator=g) / math.sqrt(D)).bfloat16()
    lens = torch.tensor(seq_kv, dtype=torch.int32)
    nblk = (lens + block_size - 1) // block_size
    total = int(nblk.sum()) * 2
    kv = torch.randn(total, 2, block_size, hkv, D, generator=g).bfloat16()
    kv[:, 0] /= math.sqrt(D)
    perm = torch.randperm(total, generator=g)[: int(nblk.sum())].to(torch.int32)
    bid = torch.zeros(len(seq_q), int(nblk.max()), dtype=torch.int32)
    o = 0
    for i in range(len(seq_q)):
        bid[i, : int(nblk[i])] = perm[o : o + int(nblk[i])]
        o += int(nblk[i])
    cu = torch.tensor([0] + list(torch.tensor(seq_q).cumsum(0)), dtype=torch.int32)
    return q, kv, cu, bid, lens


def test_oracle_prefill_bf16_matches_decode_oracle_at_one_token():
    q, kv, cu, bid, lens = paged_case([1, 1], [200, 64], 8, 1, 64)
    out = oattn.ref_prefill_bf16(q, kv[:, 0], kv[:, 1], cu, bid, lens)
    dec = oattn.ref_attn_with_paged_kvcache(q, kv, bid, (lens + 63) // 64, 1, lens - 1)
    assert allclose(dec, out, atol=1e-2, rtol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("num_batch", [1, 4])
@pytest.mark.parametrize("num_seq_q", [100, 500, 1500])
@pytest.mark.parametrize("num_seq_kv", [1500, 3000])
@pytest.mark.parametrize("kv_layout", ["nhd", "hnd"])
def test_attention_with_kvcache_prefill_bf16(num_batch, num_seq_q, num_seq_kv, kv_layout):
    import hpc

    if kv_layout == "hnd" and (num_batch != 4 or num_seq_q == 100):
        pytest.skip("HND sampled")
    q, kv, cu, bid, lens = paged_case([num_seq_q] * num_batch, [num_seq_kv] * num_batch, 4, 1, 64)
    gt = oattn.ref_prefill_bf16(q, kv[:, 0], kv[:, 1], cu, bid, lens)
    kvd = kv.cuda()
    kc, vc = kvd[:, 0], kvd[:, 1]
    if kv_layout == "hnd":
        kc = kc.transpose(1, 2).contiguous().transpose(1, 2)
        vc = vc.transpose(1, 2).contiguous().transpose(1, 2)
    out = torch.empty(q.shape, dtype=torch.bfloat16, device="cuda")
    my = hpc.attention_with_kvcache_prefill_bf16(q.cuda(), kc, vc, cu.cuda(), bid.cuda(), lens.cuda(), num_seq_q, output=out)
    assert my.data_ptr() == out.data_ptr()
    assert allclose(gt, my.cpu(), atol=0.016, rtol=0.016)


@pytest.mark.gpu
@pytest.mark.parametrize("hq,hkv", [(8, 1), (32, 4), (16, 8), (4, 4)])
@pytest.mark.parametrize("block_size", [16, 64])
def test_prefill_bf16_paged_ragged(hq, hkv, block_size):
    import hpc

    seq_q, seq_kv = [1, 37, 128, 300, 5, 64], [1, 37, 500, 300, 1000, 65]
    q, kv, cu, bid, lens = paged_case(seq_q, seq_kv, hq, hkv, block_size, seed=7)
    gt = oattn.ref_prefill_bf16(q, kv[:, 0], kv[:, 1], cu, bid, lens)
    kvd = kv.cuda()
    my = hpc.attention_with_kvcache_prefill_bf16(q.cuda(), kvd[:, 0], kvd[:, 1], cu.cuda(), bid.cuda(), lens.cuda(),
                                                 max(seq_q))
    assert allclose(gt, my.cpu(), atol=0.016, rtol=0.016)


@pytest.mark.gpu
@pytest.mark.parametrize("seq", [[2007], [3907, 100, 1, 17, 64, 65]])
@pytest.mark.parametrize("hq,hkv", [(4, 1), (16, 2)])
@pytest.mark.parametrize("use_output", [True, False])
def test_attention_prefill_bf16_contiguous(seq, hq, hkv, use_output):
    import hpc

    g = torch.Generator().manual_seed(41)
    D, total = 128, sum(seq)
    q = (torch.randn(total, hq, D, generator=g) / math.sqrt(D)).bfloat16()
    k = (torch.randn(total, hkv, D, generator=g) / math.sqrt(D)).bfloat16()
    v = torch.randn(total, hkv, D, generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(seq).cumsum(0)), dtype=torch.int32)
    gt = oattn.ref_prefill_bf16(q, k, v, cu, None, None)
    out = torch.empty(q.shape, dtype=torch.bfloat16, device="cuda") if use_output else None
    my = hpc.attention_prefill_bf16(q.cuda(), k.cuda(), v.cuda(), torch.tensor(seq, dtype=torch.int32).cuda(), cu.cuda(),
                                    max(seq), output=out)
    assert allclose(gt, my.cpu(), atol=0.016, rtol=0.016)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("hq,hkv,block_size", [(8, 1, 64), (16, 4, 32), (32, 4, 16)])
def test_prefill_bf16_transposing_reads_equal_the_perm_form(hq, hkv, block_size):
    """Round 5: V^T operands come out of LDS through ds_read_b64_tr_b16 (swizzled unpadded V stage, output block u = dims
    16 u .. 16 u + 15) instead of 16-byte row reads + v_perm_b32 (output block jj = dims 8 i + jj; development key 46 = 1).
    Both forms feed every MFMA the same eight tokens per k-group in the same order, so every output element is the same
    sum in the same order: `torch.equal`, ragged lengths and page sizes included."""
    import hpc
    from utils import dev_set

    seq_q, seq_kv = [37, 200, 1, 129], [100, 200, 700, 129]
    q, kv, cu, bid, lens = paged_case(seq_q, seq_kv, hq, hkv, block_size)
    kvd = kv.cuda()
    args = (q.cuda(), kvd[:, 0], kvd[:, 1], cu.cuda(), bid.cuda(), lens.cuda(), max(seq_q))
    new = hpc.attention_with_kvcache_prefill_bf16(*args)
    dev_set(46, 1)
    try:
        old = hpc.attention_with_kvcache_prefill_bf16(*args)
    finally:
        dev_set(46, 0)
    torch.cuda.synchronize()
    assert torch.equal(new, old)
    gt = oattn.ref_prefill_bf16(q, kv[:, 0], kv[:, 1], cu, bid, lens)
    assert allclose(gt, new.cpu(), atol=0.016, rtol=0.016)
