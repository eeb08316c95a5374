"""Parity of the per-tensor FP8 MoE surface (fuse_moe, fuse_moe_pertensor_fp8, count_and_gather,
group_gemm_pertensor_fp8, act_mul_and_quant, scaled_fp8_quant) with the CPU oracle; generators and
tolerances follow reference tests/test_fuse_moe_pertensor.py:162-223 (rtol 0.08, atol 0.1),
tests/test_fuse_moe_cp_async.py:145-242 (I = 192, H = 4096), tests/test_group_gemm_pertensor.py:52-77
and tests/test_act.py:45-59."""
import pytest
import torch

from utils import allclose, dev_set

F8 = torch.float8_e4m3fn


@pytest.mark.gpu
@pytest.mark.parametrize("num_seq,hidden,inter,num_expert", [(128, 512, 512, 128), (64, 4096, 192, 192),
                                                             (5, 512, 256, 16)])
@pytest.mark.parametrize("rank_ep,size_ep", [(0, 1), (1, 4)])
@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("use_bf16_mul", [False, True])
def test_fuse_moe_pertensor(num_seq, hidden, inter, num_expert, rank_ep, size_ep, shared, use_bf16_mul):
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(41)
    k = 8
    ids = torch.sort(torch.randint(0, num_expert, (num_seq, k), dtype=torch.int32), dim=1)[0]
    x = (torch.randn((num_seq, hidden)) / 100).to(F8)
    el = num_expert // size_ep
    guw = torch.randn((el, inter * 2, hidden)).to(F8)
    dw = torch.randn((el, hidden, inter)).to(F8)
    gus, ds, ams = torch.randn(el), torch.randn(el), torch.randn(1)
    sc = torch.randn((num_seq, k)) / k
    so = torch.randn((num_seq, hidden), dtype=torch.bfloat16) if shared else None
    gt = omoe.fuse_moe_pertensor_fp8(x, guw, dw, gus, ds, ams, ids, sc, rank_ep, so, use_bf16_mul)
    c = lambda t: None if t is None else t.cuda()  # noqa: E731
    my = hpc.fuse_moe_pertensor_fp8(c(x), c(guw), c(dw), c(gus), c(ds), c(ams), c(ids), c(sc), rank_ep, el,
                                    use_bf16_mul=use_bf16_mul, shared_output=c(so))
    out = torch.empty_like(my)
    my2 = hpc.fuse_moe(c(x), c(guw), c(dw), c(gus), c(ds), c(ams), c(ids), c(sc), rank_ep, el,
                       use_bf16_mul=use_bf16_mul, shared_output=c(so), output=out)
    torch.cuda.synchronize()
    assert allclose(gt.float(), my.cpu().float(), rtol=0.08, atol=0.1)
    assert my2.data_ptr() == out.data_ptr() and torch.equal(my2, my)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("use_bf16_mul", [False, True])
@pytest.mark.parametrize("num_seq,hidden,inter,num_expert,topk", [(600, 512, 256, 4, 2), (1500, 512, 128, 8, 4), (257, 1024, 384, 2, 2)])
def test_fuse_moe_pertensor_activation_epilogue(use_bf16_mul, num_seq, hidden, inter, num_expert, topk):
    """Long groups (>= 192 rows per expert): the gate-up GEMM of the per-tensor fused op - the API the reference's
    benchmark driver calls, benchmark/fused_moe/backends/hpcops.py:49-57 - runs on the 256 x 256 tile kernel with
    silu(gate) * up * scale -> e4m3 in its epilogue.  Bit-equal to the same GEMM followed by the separate activation
    kernel (development key 19 = 1) and within the reference tolerance of the oracle."""
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(11)
    ids = torch.sort(torch.multinomial(torch.ones(num_seq, num_expert), topk, replacement=False).to(torch.int32), dim=1)[0]
    assert num_seq * topk // num_expert >= 192
    x = (torch.randn((num_seq, hidden)) / 100).to(F8)
    guw = torch.randn((num_expert, inter * 2, hidden)).to(F8)
    dw = torch.randn((num_expert, hidden, inter)).to(F8)
    gus, ds, ams = torch.rand(num_expert) + 0.5, torch.rand(num_expert) + 0.5, torch.rand(1) + 0.5
    sc = torch.rand((num_seq, topk)) / topk
    gt = omoe.fuse_moe_pertensor_fp8(x, guw, dw, gus, ds, ams, ids, sc, 0, None, use_bf16_mul)
    c = lambda t: t.cuda()  # noqa: E731
    run = lambda: hpc.fuse_moe_pertensor_fp8(c(x), c(guw), c(dw), c(gus), c(ds), c(ams), c(ids), c(sc), 0, num_expert,  # noqa: E731
                                             use_bf16_mul=use_bf16_mul)
    fused = run()
    dev_set(19, 1)
    try:
        apart = run()
        torch.cuda.synchronize()
    finally:
        dev_set(19, 0)
    assert torch.equal(fused, apart)
    assert allclose(gt.float(), fused.cpu().float(), rtol=0.08, atol=0.1)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("use_bf16_mul", [False, True])
@pytest.mark.parametrize("num_seq,hidden,inter,num_expert,topk", [(530, 512, 256, 4, 2), (300, 1088, 128, 2, 2),  # 1088: K % 128 == 64
                                                                  (1050, 1024, 384, 4, 2), (2100, 512, 256, 8, 2)])
def test_fuse_moe_pertensor_tail_body_is_bit_identical(use_bf16_mul, num_seq, hidden, inter, num_expert, topk):
    """Per-tensor fused op with experts that end in a short tail: the tail body of the 256 x 256 kernel (gate-up GEMM with
    silu(gate) * up * scale -> e4m3 in its epilogue, down GEMM; K % 128 == 64 through the k-tail instantiation) against
    the round-4 dispatch (development key 21 = 2: tails on the half-tile body), bit for bit."""
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(num_seq)
    ids = torch.sort(torch.multinomial(torch.ones(num_seq, num_expert), topk, replacement=False).to(torch.int32), dim=1)[0]
    x = (torch.randn((num_seq, hidden)) / 100).to(F8)
    guw = torch.randn((num_expert, inter * 2, hidden)).to(F8)
    dw = torch.randn((num_expert, hidden, inter)).to(F8)
    gus, ds, ams = torch.rand(num_expert) + 0.5, torch.rand(num_expert) + 0.5, torch.rand(1) + 0.5
    sc = torch.rand((num_seq, topk)) / topk
    gt = omoe.fuse_moe_pertensor_fp8(x, guw, dw, gus, ds, ams, ids, sc, 0, None, use_bf16_mul)
    c = lambda t: t.cuda()  # noqa: E731
    run = lambda: hpc.fuse_moe_pertensor_fp8(c(x), c(guw), c(dw), c(gus), c(ds), c(ams), c(ids), c(sc), 0, num_expert,  # noqa: E731
                                             use_bf16_mul=use_bf16_mul).cpu()
    outs = {}
    dev_set(3, 4)
    try:
        # half-tile body / tail body (the product) / register-streamed tail body (development variant) / "r5": development key
        # 49 = 1 - no ride-along rows (round 6: a short tail rides along with the group's full tiles in the per-tensor kernels too;
        # the K % 128 == 64 instantiation has none)
        for key in (2, 0, "regs", "r5"):
            dev_set(21, 0 if key in ("regs", "r5") else key)
            dev_set(26, 1 if key == "regs" else 0)
            dev_set(49, 1 if key == "r5" else 0)
            outs[key] = run()
    finally:
        dev_set(21, 0)
        dev_set(26, 0)
        dev_set(49, 0)
        dev_set(3, 0)
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs["regs"]) and torch.equal(outs[0], outs["r5"])
    assert allclose(gt.float(), outs[0].float(), rtol=0.08, atol=0.1)


@pytest.mark.gpu
def test_count_and_gather_bit_exact():
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(2)
    T, k, E, rank, el, H = 77, 8, 64, 1, 16, 256
    ids = torch.sort(torch.randint(0, E, (T, k), dtype=torch.int32), dim=1)[0]
    x = torch.randn(T, H).to(F8)
    xg_ref, _, pos_ref, cnt_ref, cu_ref = omoe.gather_expert_inputs(x, torch.zeros(T, 1), ids, el, rank)
    xg, g_out, pos, seqlens, cu, tiles, cu_tiles, _, _ = hpc.count_and_gather(x.cuda(), ids.cuda(), el, rank, 512, 8)
    torch.cuda.synchronize()
    assert g_out.shape == (T * k, 512) and g_out.dtype == torch.bfloat16
    assert torch.equal(pos.cpu(), pos_ref) and torch.equal(seqlens.cpu(), cnt_ref) and torch.equal(cu.cpu(), cu_ref)
    total = int(cu_ref[-1])
    assert torch.equal(xg.cpu().view(torch.uint8)[:total], xg_ref.view(torch.uint8)[:total])
    assert torch.equal(tiles.cpu(), (cnt_ref + 7) // 8)


@pytest.mark.gpu
@pytest.mark.parametrize("actual_m", [8, 30, 70])
def test_group_gemm_pertensor(actual_m):
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(0)
    G, n, k = 8, 1024, 1792
    seqlens = torch.full((G,), actual_m, dtype=torch.int32)
    seqlens[2] = 0
    total = int(seqlens.sum())
    x = torch.randn((total, k)).to(F8)
    w = torch.randn((G, n, k)).to(F8)
    scale = torch.rand(G) + 0.5
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
    gt = omoe.group_gemm_pertensor(x, w, seqlens, cu, scale)
    my = hpc.group_gemm_pertensor_fp8(x.cuda(), w.cuda(), seqlens.cuda(), cu.cuda(), scale.cuda(),
                                      num_seq_per_group_avg=actual_m)
    torch.cuda.synchronize()
    assert allclose(gt.float(), my.cpu().float(), rtol=0.08, atol=0.5)  # |y| ~ 40: bf16 rounding


@pytest.mark.gpu
@pytest.mark.parametrize("use_bf16_mul", [True, False])
def test_act_mul_and_quant(use_bf16_mul):
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(0)
    gate_up = torch.randn((513, 4608 * 2), dtype=torch.bfloat16)
    scale = torch.rand(1) + 1.0
    gt = omoe.act_mul_and_quant(gate_up, scale, use_bf16_mul)
    out = hpc.act_mul_and_quant(gate_up.cuda(), scale.cuda(), use_bf16_mul=use_bf16_mul)
    torch.cuda.synchronize()
    # fp8 results: allow one e4m3 rounding step (exp / reciprocal differ by an ulp from torch's)
    assert allclose(gt.float(), out.cpu().float(), rtol=0.13, atol=2e-3)
    agree = (gt.view(torch.uint8) == out.cpu().view(torch.uint8)).float().mean().item()
    assert agree > 0.995, agree


@pytest.mark.gpu
def test_act_mul_and_quant_any_rank_and_output_checks():
    """reference act_mul_and_quant_entry (src/activation/entry.cc:17-47): rows = product of the leading dims, output =
    the input's shape with the last dim halved; a caller's `output` of the wrong dtype / size is refused (ADVICE round 4:
    the 2-D limit and the unchecked write-through were carried over from the old Python entry)."""
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(1)
    gate_up = torch.randn((3, 5, 2 * 256), dtype=torch.bfloat16)
    scale = torch.rand(1) + 1.0
    gt = omoe.act_mul_and_quant(gate_up.reshape(15, 512), scale, True).reshape(3, 5, 256)
    out = hpc.act_mul_and_quant(gate_up.cuda(), scale.cuda(), use_bf16_mul=True)
    assert tuple(out.shape) == (3, 5, 256) and out.dtype == torch.float8_e4m3fn
    assert (gt.view(torch.uint8) == out.cpu().view(torch.uint8)).float().mean().item() > 0.995
    mine = torch.empty((3, 5, 256), dtype=torch.float8_e4m3fn, device="cuda")
    again = hpc.act_mul_and_quant(gate_up.cuda(), scale.cuda(), use_bf16_mul=True, output=mine)
    assert again.data_ptr() == mine.data_ptr() and torch.equal(mine.view(torch.uint8), out.view(torch.uint8))
    one_d = hpc.act_mul_and_quant(gate_up[0, 0].cuda(), scale.cuda())
    assert tuple(one_d.shape) == (256,) and torch.equal(one_d.view(torch.uint8), out[0, 0].view(torch.uint8))
    with pytest.raises(RuntimeError, match="last dim halved"):
        hpc.act_mul_and_quant(gate_up.cuda(), scale.cuda(), output=torch.empty((3, 5, 128), dtype=torch.float8_e4m3fn, device="cuda"))
    with pytest.raises(RuntimeError, match="float8_e4m3fn"):
        hpc.act_mul_and_quant(gate_up.cuda(), scale.cuda(), output=torch.empty((3, 5, 256), dtype=torch.uint8, device="cuda"))
    with pytest.raises(RuntimeError, match="at least one element"):
        hpc.act_mul_and_quant(gate_up.cuda(), torch.empty(0, device="cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(513, 4608), (37, 123), (1, 1), (7,), (3, 5, 64)])
def test_scaled_fp8_quant_vs_oracle(dtype, shape):
    """hpc.scaled_fp8_quant against oracle/fuse_moe.py::scaled_fp8_quant (the restatement of the reference kernel,
    src/activation/activation.cu:461-505, pinned in tests/golden/make_golden.py): output = e4m3(input * (1 / scale)),
    saturating, fp32 / fp16 / bf16 inputs, any numel (vector body + ragged tail), returns (output, scale)."""
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(3)
    x = (torch.randn(shape) * 3.0).to(dtype)          # |x / 1e-2| reaches ~1000: the saturating conversion is covered
    if x.numel() > 16:
        x.view(-1)[5] = float("nan")
        x.view(-1)[6] = -0.0
    for sc in (torch.full((), 1e-2), torch.tensor([0.37]), torch.tensor([[3.0]])):
        want, _ = omoe.scaled_fp8_quant(x, sc)
        scd = sc.cuda()
        got, got_scale = hpc.scaled_fp8_quant(x.cuda(), scd)
        assert got_scale is scd or got_scale.data_ptr() == scd.data_ptr()      # the entry hands the scale tensor back
        assert got.shape == x.shape and got.dtype == F8
        assert torch.equal(got.cpu().view(torch.uint8), want.view(torch.uint8))
    # caller-provided output is used and returned
    out = torch.zeros(shape, dtype=F8, device="cuda")
    got, _ = hpc.scaled_fp8_quant(x.cuda(), torch.tensor([0.5]).cuda(), out)
    assert got.data_ptr() == out.data_ptr()
    assert torch.equal(out.cpu().view(torch.uint8), omoe.scaled_fp8_quant(x, torch.tensor([0.5]))[0].view(torch.uint8))


@pytest.mark.gpu
def test_scaled_fp8_quant_reference_benchmark_call_and_errors():
    """The reference benchmark driver's call (benchmark/fused_moe/backends/hpcops.py:29,49-51): fp16 activations, a
    0-dim scale of 1e-2, `torch.ops.hpc.scaled_fp8_quant(x, scale, None)` unpacked into two values; and the entry's
    checks (src/activation/entry.cc:163-186)."""
    import hpc  # noqa: F401
    from oracle import fuse_moe as omoe

    torch.manual_seed(0)
    a_half = torch.randn(64, 4096, dtype=torch.half)
    a_scale = torch.full((), 1e-2, dtype=torch.float32)
    x_fp8, _ = torch.ops.hpc.scaled_fp8_quant(a_half.cuda(), a_scale.cuda(), None)
    assert torch.equal(x_fp8.cpu().view(torch.uint8), omoe.scaled_fp8_quant(a_half, a_scale)[0].view(torch.uint8))
    # = 100 * a (not 0.01 * a): dequantised values are near a / scale
    assert allclose(x_fp8.cpu().float(), (a_half.float() * 100).clamp(-448, 448), rtol=0.07, atol=0.02)
    xc, sc = a_half.cuda(), a_scale.cuda()
    with pytest.raises(RuntimeError, match="scale is required"):
        torch.ops.hpc.scaled_fp8_quant(xc, None, None)
    with pytest.raises(RuntimeError, match="float32, float16, or bfloat16"):
        torch.ops.hpc.scaled_fp8_quant(xc.to(torch.float64), sc, None)
    with pytest.raises(RuntimeError, match="non-empty"):
        torch.ops.hpc.scaled_fp8_quant(xc[:0], sc, None)
    with pytest.raises(RuntimeError, match="contiguous"):
        torch.ops.hpc.scaled_fp8_quant(xc.t(), sc, None)
    with pytest.raises(RuntimeError, match="one element"):
        torch.ops.hpc.scaled_fp8_quant(xc, torch.ones(2, device="cuda"), None)
    with pytest.raises(RuntimeError, match="scale dtype must be float32"):
        torch.ops.hpc.scaled_fp8_quant(xc, sc.half(), None)
    with pytest.raises(RuntimeError, match="output shape must match"):
        torch.ops.hpc.scaled_fp8_quant(xc, sc, torch.empty(3, dtype=F8, device="cuda"))


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("tiled_mode", [2, 3, 4, 12, 22])  # 12 = LDS-DMA ring kernel with 32-token tiles; 4 = 256x256 kernel
@pytest.mark.parametrize("k", [1792, 1088, 192])  # 1088 and 192 end in a half k-block (K % 128 == 64)
def test_group_gemm_pertensor_tiled_kernels(tiled_mode, k):
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(2)
    n = 512
    seqlens = torch.tensor([200, 0, 129, 3, 260], dtype=torch.int32)
    G, total = len(seqlens), int(seqlens.sum())
    x = torch.randn((total, k)).to(F8)
    w = torch.randn((G, n, k)).to(F8)
    scale = torch.rand(G) + 0.5
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
    gt = omoe.group_gemm_pertensor(x, w, seqlens, cu, scale)
    dev_set(3, tiled_mode % 10)
    dev_set(6, 1 + tiled_mode // 10)  # 1x: 32-token tiles, 2x: 64-token tiles
    try:
        my = hpc.group_gemm_pertensor_fp8(x.cuda(), w.cuda(), seqlens.cuda(), cu.cuda(), scale.cuda(),
                                          num_seq_per_group_avg=total // G)
        torch.cuda.synchronize()
    finally:
        dev_set(3, 0)
        dev_set(6, 0)
    assert allclose(gt.float(), my.cpu().float(), rtol=0.08, atol=0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("num_group,actual_m,n,k", [(48, 64, 512, 256), (48, 42, 512, 192), (16, 8, 1024, 256),
                                                    (8, 200, 256, 1024)])
@pytest.mark.parametrize("scatter", [False, True])
def test_group_gemm_cp_async_ops(num_group, actual_m, n, k, scatter):
    """raw torch.ops.hpc.group_gemm_fp8[_scatter]_cp_async (reference tests/test_group_gemm_cp_async.py:65-110):
    the scatter form reads its rows from a pool through row_indices."""
    import hpc  # noqa: F401
    from oracle import fuse_moe as omoe

    g = torch.Generator().manual_seed(10086)
    seqlens = torch.full((num_group,), actual_m, dtype=torch.int32)
    seqlens[1] = max(actual_m - 5, 0)
    total = int(seqlens.sum())
    pool = torch.randn((total + 7, k), generator=g).to(F8)
    w = torch.randn((num_group, n, k), generator=g).to(F8)
    scale = torch.rand(num_group, generator=g) + 0.5
    rows = torch.randperm(total + 7, generator=g)[:total].to(torch.int32)
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
    tiles = (seqlens + 63) // 64
    cu_tiles = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(tiles, 0).to(torch.int32)])
    x_compact = pool[rows.long()]
    gt = omoe.group_gemm_pertensor(x_compact, w, seqlens, cu, scale)
    d = lambda t: t.cuda()  # noqa: E731
    if scatter:
        my = torch.ops.hpc.group_gemm_fp8_scatter_cp_async(d(pool), d(w), d(scale), d(rows), d(seqlens), d(cu),
                                                           d(tiles), d(cu_tiles), False)
    else:
        my = torch.ops.hpc.group_gemm_fp8_cp_async(d(x_compact), d(w), d(scale), d(seqlens), d(cu), d(tiles),
                                                   d(cu_tiles), True)
    assert allclose(gt.float(), my.cpu().float(), rtol=0.08, atol=1)


@pytest.mark.gpu
def test_masked_act_variants():
    """masked_act_mul_and_quant / masked_act_mul_and_blockwise_quant (reference tests/test_act.py:62-160):
    valid rows match the oracle, rows past num_per_expert[e] keep whatever the output held."""
    import hpc
    from oracle import fuse_moe as omoe

    g = torch.Generator().manual_seed(41)
    E, per, inter = 32, 48, 2048
    total = E * per
    gate_up = torch.randn((total, 2 * inter), generator=g).bfloat16()
    scale = torch.randn(1, generator=g)
    cnt = torch.randint(0, per, (E,), generator=g).to(torch.int32)
    keep = (torch.arange(total) % per) < cnt[torch.arange(total) // per]
    sentinel = torch.full((total, inter), 3.0).to(F8)
    out = sentinel.clone().cuda()
    my = hpc.masked_act_mul_and_quant(gate_up.cuda(), scale.cuda(), cnt.cuda(), output=out)
    gt = omoe.act_mul_and_quant(gate_up, scale, use_bf16_mul=False)
    assert my.data_ptr() == out.data_ptr()
    assert allclose(gt.float()[keep], my.cpu().float()[keep], atol=0.15, rtol=0.13)
    assert torch.equal(my.cpu().view(torch.uint8)[~keep], sentinel.view(torch.uint8)[~keep])
    qb, sb = hpc.masked_act_mul_and_blockwise_quant(gate_up.cuda(), cnt.cuda())
    gq, gs = omoe.act_mul_and_blockwise_quant(gate_up)
    assert qb.dtype == F8 and sb.shape == (total, inter // 128)
    assert allclose(gs[keep], sb.cpu()[keep], rtol=1e-3, atol=1e-6)
    deq_my = qb.cpu().float()[keep] * sb.cpu()[keep].repeat_interleave(128, dim=1)
    deq_gt = gq.float()[keep] * gs[keep].repeat_interleave(128, dim=1)
    assert allclose(deq_gt, deq_my, rtol=0.13, atol=2e-3)
