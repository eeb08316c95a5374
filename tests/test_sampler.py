"""fused_sampler parity (grid of reference tests/test_sampler.py:167-625): with injected Gumbel noise the
sampled tokens must EQUAL the oracle's (oracle/sampler.py, CPU) - integer outputs, bit-exact bar."""
import pytest
import torch

from oracle import sampler as orc

V0 = 120832  # the reference's vocabulary


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def test_oracle_sampler_basics():
    """CPU: top-1 without noise is the arg-max; stable top-k ranks equal values by token id."""
    g = _gen(0)
    logits = torch.randn(3, 4096, generator=g)
    tok, _ = orc.ref_fused_sampler(logits, topk=1, gumbel_noise=torch.zeros_like(logits))
    assert torch.equal(tok.view(-1).long(), logits.argmax(-1))
    row = torch.tensor([1.0, 3.0, 3.0, 2.0, 3.0])
    assert orc.stable_topk(row, 3)[1].tolist() == [1, 2, 4]
    t = orc.ref_temperature_sample(logits, torch.full((3,), 0.5), torch.zeros_like(logits),
                                   logits.argmax(-1))  # masking the arg-max moves the sample
    assert not torch.equal(t.view(-1).long(), logits.argmax(-1))


@pytest.mark.gpu
@pytest.mark.parametrize("vocab_size", [V0, 4096, 151936])
@pytest.mark.parametrize("batch_size", [1, 4])
def test_only_logits(batch_size, vocab_size):
    import hpc

    g = _gen(0)
    logits = torch.randn(batch_size, vocab_size, generator=g)
    gumbel = orc.gumbel0_like(logits, g)
    tok = hpc.fused_sampler(logits.cuda(), gumbel_noise=gumbel.cuda())
    assert tok.shape == (batch_size, 1) and tok.dtype == torch.int32
    ref, _ = orc.ref_fused_sampler(logits, gumbel_noise=gumbel, max_topk=32)
    assert torch.equal(tok.cpu(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("batch_size", [1, 4])
@pytest.mark.parametrize("max_topk,topk_val", [(32, 20), (64, 50)])
@pytest.mark.parametrize("softmax_policy,topp_val", [(0, 0.0), (1, 0.9), (2, 0.9), (2, 0.3)])
def test_topk_topp_softmax_matrix(batch_size, max_topk, topk_val, softmax_policy, topp_val):
    import hpc

    g = _gen(1234 + max_topk + topk_val + int(topp_val * 10))
    logits = torch.randn(batch_size, V0, generator=g)
    gumbel = orc.gumbel0_like(logits, g)
    topk_t = torch.full((batch_size,), topk_val, dtype=torch.int32)
    topp_t = torch.full((batch_size,), topp_val) if topp_val > 0 else 0.0
    tok = hpc.fused_sampler(logits.cuda(), softmax_policy=hpc.SoftmaxPolicy(softmax_policy), topk=topk_t.cuda(),
                            topp=topp_t.cuda() if topp_val > 0 else 0.0, max_topk=max_topk,
                            gumbel_noise=gumbel.cuda())
    ref, _ = orc.ref_fused_sampler(logits, softmax_policy=softmax_policy, topk=topk_t, topp=topp_t,
                                   max_topk=max_topk, gumbel_noise=gumbel)
    assert torch.equal(tok.cpu(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("softmax_policy", [1, 2])
def test_fully_masked_segments(softmax_policy):
    """grammar / vocabulary masks: whole segments of the row are -inf.  With softmax BEFORE top-k a segment's
    (max, sum) statistics must come out as (-inf, 0) - not exp(-inf - -inf) = NaN, which used to turn the whole
    row into 'smallest token id of the top-k'."""
    import hpc

    g = _gen(77)
    logits = torch.randn(3, V0, generator=g)
    logits[0, : V0 // 2] = float("-inf")          # leading segments fully masked
    logits[1, V0 // 3:] = float("-inf")           # trailing segments fully masked
    logits[2, 1000: V0 - 1000] = float("-inf")    # only the ends survive
    gumbel = orc.gumbel0_like(logits, g)
    topk_t = torch.full((3,), 20, dtype=torch.int32)
    topp_t = torch.full((3,), 0.9)
    tok = hpc.fused_sampler(logits.cuda(), softmax_policy=hpc.SoftmaxPolicy(softmax_policy), topk=topk_t.cuda(),
                            topp=topp_t.cuda(), max_topk=32, gumbel_noise=gumbel.cuda())
    ref, _ = orc.ref_fused_sampler(logits, softmax_policy=softmax_policy, topk=topk_t, topp=topp_t, max_topk=32,
                                   gumbel_noise=gumbel)
    assert torch.equal(tok.cpu(), ref)
    assert bool(torch.isfinite(logits[torch.arange(3), tok.cpu().long().flatten()]).all())  # never samples a masked token


@pytest.mark.gpu
@pytest.mark.parametrize("batch_size", [1, 4])
def test_repetition_penalty_and_writeback(batch_size):
    import hpc

    g = _gen(7)
    logits = torch.randn(batch_size, V0, generator=g)
    gumbel = orc.gumbel0_like(logits, g)
    max_bs = batch_size + 3
    penalty = torch.randint(0, 256, (max_bs, (V0 + 7) // 8), generator=g).to(torch.uint8)
    slot_id = torch.randperm(max_bs, generator=g)[:batch_size].to(torch.int32)
    kw = dict(repetition_penalty=1.05, temperature=0.7, max_topk=32)
    topk_t = torch.full((batch_size,), 20, dtype=torch.int32)
    topp_t = torch.full((batch_size,), 0.9)
    ref_tok, ref_pen = orc.ref_fused_sampler(logits, penalty_mask=penalty.clone(), slot_id=slot_id, softmax_policy=2,
                                             topk=topk_t, topp=topp_t, gumbel_noise=gumbel, **kw)
    pen_dev = penalty.cuda()
    tok = hpc.fused_sampler(logits.cuda(), penalty_mask=pen_dev, slot_id=slot_id.cuda(),
                            softmax_policy=hpc.SoftmaxPolicy.AFTER_TOPK, topk=topk_t.cuda(), topp=topp_t.cuda(),
                            gumbel_noise=gumbel.cuda(), **kw)
    assert torch.equal(tok.cpu(), ref_tok)
    assert torch.equal(pen_dev.cpu(), ref_pen), "penalty writeback mismatch"


@pytest.mark.gpu
def test_bf16_logits_and_value_ties():
    """bf16 logits: dozens of equal values inside the top-k - the stable (smaller token id first) rule decides."""
    import hpc

    g = _gen(3)
    logits = torch.randn(2, V0, generator=g).bfloat16()
    gumbel = orc.gumbel0_like(logits, g)
    for pol, tp in ((2, 0.9), (0, 0.0), (1, 0.95)):
        tok = hpc.fused_sampler(logits.cuda(), softmax_policy=hpc.SoftmaxPolicy(pol), topk=20, topp=tp, max_topk=32,
                                gumbel_noise=gumbel.cuda())
        ref, _ = orc.ref_fused_sampler(logits.float(), softmax_policy=pol, topk=20, topp=tp, max_topk=32,
                                       gumbel_noise=gumbel)
        assert torch.equal(tok.cpu(), ref), (pol, tok.flatten(), ref.flatten())
    # a row of few distinct values: the top-64 is decided almost entirely by the tie rule
    coarse = (torch.randint(0, 4, (3, 8192), generator=g).float() - 2).bfloat16()
    gum = orc.gumbel0_like(coarse, g)
    tok = hpc.fused_sampler(coarse.cuda(), topk=64, max_topk=64, gumbel_noise=gum.cuda())
    ref, _ = orc.ref_fused_sampler(coarse.float(), topk=64, max_topk=64, gumbel_noise=gum)
    assert torch.equal(tok.cpu(), ref)


@pytest.mark.gpu
def test_scalar_vs_tensor_equivalence():
    import hpc

    g = _gen(5)
    B = 4
    logits = torch.randn(B, V0, generator=g).cuda()
    gumbel = orc.gumbel0_like(logits.cpu(), g).cuda()
    t1 = hpc.fused_sampler(logits, temperature=0.7, topk=20, max_topk=32, softmax_policy=hpc.SoftmaxPolicy.AFTER_TOPK,
                           topp=0.9, gumbel_noise=gumbel)
    t2 = hpc.fused_sampler(logits, temperature=torch.full((B,), 0.7, device="cuda"),
                           topk=torch.full((B,), 20, dtype=torch.int64, device="cuda"), max_topk=32,
                           softmax_policy=2, topp=torch.full((B,), 0.9, device="cuda"), gumbel_noise=gumbel)
    assert torch.equal(t1, t2)


@pytest.mark.gpu
def test_own_noise_smoke_and_freshness():
    """No injected noise: Philox draws; tokens stay inside the top-k and differ from launch to launch."""
    import hpc

    g = _gen(9)
    logits = torch.randn(2, V0, generator=g).cuda()
    top20 = torch.topk(logits, k=20, dim=-1).indices
    seen = set()
    for _ in range(16):
        tok = hpc.fused_sampler(logits, topk=20, max_topk=32, seed=42)
        assert tok.shape == (2, 1) and tok.dtype == torch.int32
        for b in range(2):
            assert int(tok[b, 0]) in top20[b].tolist()
        seen.add(tuple(tok.flatten().tolist()))
    assert len(seen) > 1


@pytest.mark.gpu
@pytest.mark.parametrize("pad", [4, 256])
def test_padded_logits_stride(pad):
    import hpc

    g = _gen(17)
    B = 4
    padded = torch.randn(B, V0 + pad, generator=g)
    padded[:, V0:] = float("inf")
    dev = padded.cuda()[:, :V0]
    assert dev.stride(0) == V0 + pad and not dev.is_contiguous()
    gumbel = orc.gumbel0_like(padded[:, :V0], g)
    tok = hpc.fused_sampler(dev, softmax_policy=2, topk=20, topp=0.9, max_topk=32, gumbel_noise=gumbel.cuda())
    ref, _ = orc.ref_fused_sampler(padded[:, :V0].contiguous(), softmax_policy=2, topk=20, topp=0.9, max_topk=32,
                                   gumbel_noise=gumbel)
    assert torch.equal(tok.cpu(), ref)


@pytest.mark.gpu
def test_sampler_error_paths():
    import hpc

    logits = torch.randn(1, V0, device="cuda")
    penalty = torch.zeros((4, (V0 + 7) // 8), dtype=torch.uint8, device="cuda")
    with pytest.raises((RuntimeError, ValueError)):
        hpc.fused_sampler(logits, softmax_policy=2, topp=0.9, seed=1)  # topp without topk
    with pytest.raises((RuntimeError, ValueError)):
        hpc.fused_sampler(logits, topk=20, topp=0.9, max_topk=32, seed=1)  # topp without softmax
    with pytest.raises((RuntimeError, ValueError)):
        hpc.fused_sampler(logits, penalty_mask=penalty, seed=1)  # mask without slot_id
    with pytest.raises((RuntimeError, ValueError)):
        hpc.fused_sampler(torch.randn(1, 12345, device="cuda"), seed=42)  # unsupported vocab
    with pytest.raises((RuntimeError, ValueError)):
        hpc.fused_sampler(logits, topk=20, max_topk=16, seed=1)
    with pytest.raises((RuntimeError, ValueError)):
        hpc.fused_sampler(torch.randn(2, V0 * 2, device="cuda")[:, ::2], seed=1)  # inner stride != 1
    with pytest.raises((RuntimeError, ValueError)):
        hpc.fused_sampler(logits)  # no noise and no seed
    with pytest.raises(ValueError):
        hpc.fused_sampler(logits, topk=20, draft_token_ids=torch.zeros(1, dtype=torch.int64, device="cuda"), seed=1)


# ---------------------------------------------------------------------------- temperature fast path
@pytest.mark.gpu
@pytest.mark.parametrize("batch_size", [1, 8])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("temp_mode", ["scalar", "tensor"])
@pytest.mark.parametrize("stride_mode", ["compact", "padded"])
def test_temperature_only_golden(batch_size, dtype, temp_mode, stride_mode):
    import hpc

    g = _gen(0xC0FFEE + batch_size * 17)
    pad = 128 if stride_mode == "padded" else 0
    big = torch.randn(batch_size, V0 + pad, generator=g).to(dtype)
    logits = big[:, :V0]
    dev = big.cuda()[:, :V0]
    if temp_mode == "scalar":
        t_arg, t_ref = 0.7, torch.full((batch_size,), 0.7)
    else:
        t_ref = torch.rand(batch_size, generator=g) * 1.5 + 0.3
        t_arg = t_ref.cuda()
    gumbel = orc.gumbel0_like(logits, g)
    out = hpc.fused_sampler(dev, temperature=t_arg, gumbel_noise=gumbel.cuda())
    assert out.shape == (batch_size, 1) and out.dtype == torch.int32
    assert torch.equal(out.cpu(), orc.ref_temperature_sample(logits, t_ref, gumbel))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mask_case", ["all_minus_one", "single_per_row", "out_of_range_token", "mask_the_argmax"])
def test_temperature_draft_mask(dtype, mask_case):
    import hpc

    g = _gen(0xBEEF + len(mask_case))
    B = 8
    logits = torch.randn(B, V0, generator=g).to(dtype)
    temperature = torch.rand(B, generator=g) * 1.5 + 0.3
    gumbel = orc.gumbel0_like(logits, g)
    base = orc.ref_temperature_sample(logits, temperature, gumbel)
    if mask_case == "all_minus_one":
        draft = torch.full((B,), -1, dtype=torch.int64)
    elif mask_case == "single_per_row":
        draft = torch.randint(0, V0, (B,), generator=g)
        draft[1] = -1
    elif mask_case == "out_of_range_token":
        draft = torch.tensor([0, -1, V0, 100, V0 + 17, -1, 200, V0 * 2], dtype=torch.int64)
    else:
        draft = base.view(-1).long().clone()  # forces a different token in every row
    ref = orc.ref_temperature_sample(logits, temperature, gumbel, draft)
    if mask_case == "mask_the_argmax":
        assert not (ref == base).any()
    dl, dt, dg = logits.cuda(), temperature.cuda(), gumbel.cuda()
    out = hpc.fused_sampler(dl, temperature=dt, gumbel_noise=dg, draft_token_ids=draft.cuda())
    assert torch.equal(out.cpu(), ref)
    assert torch.equal(hpc.fused_sampler(dl, temperature=dt, gumbel_noise=dg).cpu(), base)  # logits untouched


@pytest.mark.gpu
def test_temperature_validation():
    import hpc

    B = 4
    logits = torch.randn(B, V0, device="cuda")
    hpc.fused_sampler(logits, temperature=1.0, seed=42)
    hpc.fused_sampler(logits, temperature=1.0, draft_token_ids=torch.full((B,), -1, dtype=torch.int64, device="cuda"),
                      seed=42)
    with pytest.raises(RuntimeError, match="draft_token_ids size must be"):
        hpc.fused_sampler(logits, temperature=1.0,
                          draft_token_ids=torch.full((B + 1,), -1, dtype=torch.int64, device="cuda"), seed=42)
    with pytest.raises(RuntimeError, match="draft_token_ids dtype must be int64"):
        hpc.fused_sampler(logits, temperature=1.0,
                          draft_token_ids=torch.full((B,), -1, dtype=torch.int32, device="cuda"), seed=42)
    with pytest.raises(RuntimeError):
        hpc.fused_sampler(logits, temperature=torch.tensor([1.0, 0.0, 1.0, 1.0], device="cuda"), seed=42)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_temperature_distribution(dtype):
    """Self-drawn Philox noise: empirical frequencies track softmax(logits / T)  (reference :585-625)."""
    import hpc

    g = _gen(1234)
    B, K, N = 4, 16, 2000
    top_tokens = torch.randint(0, V0, (B, K), generator=g)
    logits = torch.full((B, V0), -20.0)
    for b in range(B):
        logits[b, top_tokens[b]] = torch.randn(K, generator=g) * 2.0 + 3.0
    dev = logits.to(dtype).cuda()
    counts = torch.zeros(B, V0, device="cuda")
    for _ in range(N):
        tok = hpc.fused_sampler(dev, temperature=1.0, seed=42)
        counts.scatter_add_(1, tok.to(torch.int64), torch.ones_like(tok, dtype=torch.float32))
    tv = 0.5 * (counts.cpu() / N - torch.softmax(logits, dim=-1)).abs().sum(dim=-1)
    assert (tv < 0.1).all(), tv.tolist()
